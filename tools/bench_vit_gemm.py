"""ViT-B/16 token GEMMs (M = 8 x 1121 = 8 968 tokens, bias epilogue) on conv1x1_w128_kernel (OESS_W128_MIN_TILES=1 lets it take them)
against the kernels the dispatch picks today; interleaved rounds, separate processes per mode are not needed (the knob is read once:
run the script twice)."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip

M = 8968
for name, K, N in (("qkv", 768, 2304), ("fc1 (no GELU)", 768, 3072), ("proj", 768, 768), ("fc2", 3072, 768)):
    x = (torch.randn(1, 1, M, K, device="cuda") * 0.5).bfloat16()
    w = torch.randn(N, K, 1, 1, device="cuda") / K ** 0.5
    b = torch.randn(N, device="cuda")
    packed = hip.pack_conv_weight(w)
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.bfloat16)
    t = []
    for _ in range(3):
        for _ in range(3):
            hip.conv2d_nhwc(x, packed, b, N, 1, 1, 1, 0, 1, out=out)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            hip.conv2d_nhwc(x, packed, b, N, 1, 1, 1, 0, 1, out=out)
        e1.record(); torch.cuda.synchronize()
        t.append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * K * N
    a = statistics.median(t)
    print(f"{name:14s} {K:5d} -> {N:5d}: {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s)  checksum {float(out.float().abs().mean()):.6f}", flush=True)
