import os, sys, torch
sys.path.insert(0, "/root/repo")
from openess_amd import hip
SH = [("256->1024", 140800, 1024, 256), ("512->2048", 140800, 2048, 512), ("1024->256", 140800, 256, 1024), ("2048->512", 140800, 512, 2048),
      ("1024->2048", 140800, 2048, 1024), ("1024->2048 @35200", 35200, 2048, 1024), ("2048->512 @35200", 35200, 512, 2048), ("512->2048 @35200", 35200, 2048, 512)]
for name, M, N, K in SH:
    x = torch.randn(1, 1, M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, 1, 1, device="cuda") * 0.02
    pk = hip.pack_conv_weight(w)
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.bfloat16)
    f = lambda: hip.conv2d_nhwc(x, pk, None, N, 1, 1, 1, 0, 1, out=out)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f"{name:20s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.0f} TF/s", flush=True)
