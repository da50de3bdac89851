import sys, os, torch
sys.path.insert(0, "/root/repo")
from openess_amd import hip
M = 8968
for name, N, K in (("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)):
    a = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda") * 0.02
    wb = w.bfloat16(); bias = torch.randn(N, device="cuda")
    def vend(): return torch.addmm(bias.bfloat16(), a, wb.t())
    x4 = a.view(1, 1, M, K)   # NHWC with H=1, W=M
    pk = hip.pack_conv_weight(w.view(N, K, 1, 1))
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.bfloat16)
    def mine(): return hip.conv2d_nhwc(x4, pk, bias, N, 1, 1, 1, 0, 1, out=out)
    for tag, fn in (("hipBLASLt addmm", vend), ("oess conv1x1", mine)):
        for _ in range(5): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 50
        print(f"{name:5s} {tag:16s} M={M} N={N} K={K}: {ms*1e3:7.1f} us {2.0*M*N*K/ms/1e9:6.0f} TF/s", flush=True)
