import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
B, H, W, Cin, Cout, R, st, pad, dil = 8, 110, 160, 256, 512, 3, 1, 1, 1
x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
w = torch.randn(Cout, Cin, R, R, device="cuda") * 0.05
pk = hip.pack_conv_weight(w)
out = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
torch.cuda.synchronize()
