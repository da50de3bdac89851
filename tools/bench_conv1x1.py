"""A/B of the teacher's large 1x1 layers (M = 8 x 110 x 160): conv1x1_w128_kernel (OESS_W128_GEMM=1) against conv_fwd_dma_kernel<256,256>
(=0), raw bf16 result + BatchNorm tile statistics, interleaved rounds in one process."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip

SHAPES = [(256, 1024), (512, 2048), (1024, 2048), (2048, 512), (1024, 512), (1024, 256), (512, 1024)]
if "--short" in sys.argv:
    SHAPES = SHAPES[:2]
B, H, W = 8, 110, 160
MODES = ("1",) if "--w128-only" in sys.argv else ("0", "1")      # --w128-only: ablation builds (G128_ABL) of the w128 kernel alone
M = B * H * W
for Cin, Cout in SHAPES:
    x = (torch.randn(B, H, W, Cin, device="cuda") * 0.5).bfloat16()
    w = torch.randn(Cout, Cin, 1, 1, device="cuda") / Cin ** 0.5
    packed = hip.pack_conv_weight(w)
    out = torch.empty(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    part = torch.empty((M + 127) // 128, 2, Cout, device="cuda")
    t = {"0": [], "1": []}
    for _ in range(3):
        for mode in MODES:
            os.environ["OESS_W128_GEMM"] = mode
            for _ in range(3):
                hip.conv2d_nhwc(x, packed, None, Cout, 1, 1, 1, 0, 1, out=out, tile_stats=part)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.conv2d_nhwc(x, packed, None, Cout, 1, 1, 1, 0, 1, out=out, tile_stats=part)
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * Cin * Cout
    if len(MODES) == 1:
        b = statistics.median(t["1"])
        print(f"{Cin:5d} -> {Cout:5d}: w128 {b:7.1f} us ({fl / b / 1e6:6.0f} TF/s)", flush=True)
        continue
    a, b = statistics.median(t["0"]), statistics.median(t["1"])
    print(f"{Cin:5d} -> {Cout:5d}: 256x256 8-wave {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s)   w128 {b:7.1f} us ({fl / b / 1e6:6.0f} TF/s)   {a / b:5.2f}x", flush=True)
