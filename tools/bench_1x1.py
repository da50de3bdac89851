"""The teacher's large 1x1 layers (rule 3b of conv_fwd_impl: 256 x 256 tiles) at M = 140 800 and M = 35 200: time, rate, and a check
of the result (with tile statistics) against an fp32 matmul of the same bf16 operands on a row sample.
Environment OESS_256_BK32=1 selects the BK = 32 / 4-stage ring for A/B runs."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip
SH = [("256->1024", 140800, 1024, 256), ("512->2048", 140800, 2048, 512), ("1024->256", 140800, 256, 1024), ("2048->512", 140800, 512, 2048),
      ("1024->2048", 140800, 2048, 1024), ("1024->2048 @35200", 35200, 2048, 1024), ("2048->512 @35200", 35200, 512, 2048), ("512->2048 @35200", 35200, 2048, 512),
      ("ViT fc1 768->3072 @8968", 8968, 3072, 768), ("ragged 512->768 @70001", 70001, 768, 512)]
tot = 0.0
for name, M, N, K in SH:
    torch.manual_seed(0)
    x = torch.randn(1, 1, M, K, device="cuda").bfloat16()
    w = torch.randn(N, K, 1, 1, device="cuda") * 0.02
    pk = hip.pack_conv_weight(w)
    out = torch.empty(1, 1, M, N, device="cuda", dtype=torch.bfloat16)
    part = torch.empty(((M + 127) // 128, 2, N), dtype=torch.float32, device="cuda")
    f = lambda: hip.conv2d_nhwc(x, pk, None, N, 1, 1, 1, 0, 1, out=out, tile_stats=part)
    for _ in range(3): f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    rows = torch.cat([torch.arange(0, 300), torch.arange(M // 2, M // 2 + 300), torch.arange(M - 300, M)]).cuda()
    ref = x[0, 0, rows].float() @ w.bfloat16().float().reshape(N, K).t()
    err = float((out[0, 0, rows].float() - ref).abs().max() / ref.abs().max())
    s_ref = out[0, 0].float().sum(0)
    s_err = float((part[:, 0].sum(0) - s_ref).abs().max() / s_ref.abs().max().clamp_min(1e-6))
    tot += ms
    print(f"{name:26s} {ms*1e3:8.1f} us {2.0*M*N*K/ms/1e9:7.0f} TF/s   max rel err {err:.1e}  tile-stat sum err {s_err:.1e}", flush=True)
print(f"sum {tot*1e3:.1f} us")
