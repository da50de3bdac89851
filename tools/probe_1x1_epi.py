"""Teacher 1x1 layers (M = 140 800): ours without / with the BatchNorm tile statistics next to torch.matmul (hipBLASLt) and the
bounds of the shape (MFMA at 2.5 PF, HBM bytes at 8 TB/s)."""
import sys, torch
sys.path.insert(0, "/root/repo")
from openess_amd import hip
SH = [("256->1024", 140800, 1024, 256), ("512->2048", 140800, 2048, 512), ("1024->256", 140800, 256, 1024), ("2048->512", 140800, 512, 2048),
      ("64->256", 140800, 256, 64), ("256->64", 140800, 64, 256)]
def tm(f, n=20):
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for name, M, N, K in SH:
    x = torch.randn(1, 1, M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, 1, 1, device="cuda") * 0.02
    pk = hip.pack_conv_weight(w)
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(((M + 127) // 128, 2, N), dtype=torch.float32, device="cuda")
    t0 = tm(lambda: hip.conv2d_nhwc(x, pk, None, N, 1, 1, 1, 0, 1, out=out))
    t1 = tm(lambda: hip.conv2d_nhwc(x, pk, None, N, 1, 1, 1, 0, 1, out=out, tile_stats=part))
    a2, b2 = x.view(M, K), w.view(N, K).bfloat16().t().contiguous()
    o2 = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    t2 = tm(lambda: torch.matmul(a2, b2, out=o2))
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + M * N + N * K)
    print(f"{name:12s} plain {t0:7.1f} us  stats {t1:7.1f} us  hipBLASLt {t2:7.1f} us   bounds: mfma {fl/2.5e9:6.1f} us, hbm {by/8e6:6.1f} us", flush=True)
