import sys
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from openess_amd import engine
def cos(a, b):
    a = a.float().flatten().double(); b = b.float().flatten().double(); return float(a @ b / (a.norm() * b.norm() + 1e-30))
torch.manual_seed(0)
for shape in ((2, 512, 2, 3), (2, 256, 12, 16), (2, 64, 48, 64)):
    x32 = torch.randn(*shape, device='cuda')
    w = torch.rand(shape[1], device='cuda') + 0.5; b = torch.randn(shape[1], device='cuda')
    g = torch.randn(*shape, device='cuda')
    for tag, xin in (('bf16_cl', x32.bfloat16().contiguous(memory_format=torch.channels_last)), ('bf16_nchw', x32.bfloat16()), ('f32_cl', x32.contiguous(memory_format=torch.channels_last))):
        xi = xin.clone().requires_grad_(True); wi = w.clone().requires_grad_(True)
        rm, rv = torch.zeros_like(w), torch.ones_like(w)
        y = F.batch_norm(xi, rm, rv, wi, b, True, 0.1, 1e-5)
        y = F.relu(y)
        y.backward(g.to(y.dtype))
        xr = x32.bfloat16().float().clone().requires_grad_(True); wr = w.clone().requires_grad_(True)
        yr = F.relu(F.batch_norm(xr, torch.zeros_like(w), torch.ones_like(w), wr, b, True, 0.1, 1e-5)); yr.backward(g.bfloat16().float() if 'bf16' in tag else g)
        print(shape, tag, 'y', round(cos(y, yr), 5), 'gx', round(cos(xi.grad, xr.grad), 5), 'gw', round(cos(wi.grad, wr.grad), 5))
# conv train fn vs torch
for (Cin, Cout, k, st, pad, dil, H, W) in ((256, 512, 3, 1, 1, 1, 6, 8), (512, 11, 1, 1, 0, 1, 6, 8), (64, 64, 3, 2, 1, 1, 16, 24), (2048, 256, 3, 1, 6, 6, 6, 8)):
    x = torch.randn(2, Cin, H, W, device='cuda').bfloat16().contiguous(memory_format=torch.channels_last)
    wgt = (torch.randn(Cout, Cin, k, k, device='cuda') / (Cin * k * k) ** 0.5)
    wp = torch.nn.Parameter(wgt.clone()); xp = x.clone().requires_grad_(True)
    pw = engine.PackedWeight()
    y = engine.conv2d_train(xp, wp, None, pw, k, st, pad, dil)
    g = torch.randn_like(y.float())
    y.backward(g.to(y.dtype))
    xr = x.float().clone().requires_grad_(True); wr = wgt.bfloat16().float().clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, None, st, pad, dil); yr.backward(g.bfloat16().float())
    print('conv', (Cin, Cout, k, st, pad, dil), 'y', round(cos(y, yr), 5), 'gx', round(cos(xp.grad, xr.grad), 5), 'gw', round(cos(wp.grad, wr.grad), 5))
