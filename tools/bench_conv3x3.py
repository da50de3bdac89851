"""A/B of the teacher's dilated 3x3 layers (M = 8 x 110 x 160): conv3x3_w128_kernel (OESS_W128_CONV3=1) against conv3x3_halo_kernel<0> (=0),
raw bf16 result + BatchNorm tile statistics, interleaved rounds in one process."""
import os, statistics, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip

B, H, W = 8, 110, 160
M = B * H * W
for C, dil in ((256, 2), (512, 4), (256, 1), (512, 1)):
    x = (torch.randn(B, H, W, C, device="cuda") * 0.5).bfloat16()
    w = torch.randn(C, C, 3, 3, device="cuda") / (9 * C) ** 0.5
    packed = hip.pack_conv_weight(w)
    out = torch.empty(B, H, W, C, device="cuda", dtype=torch.bfloat16)
    part = torch.empty((M + 127) // 128, 2, C, device="cuda")
    t = {"0": [], "1": []}
    for _ in range(3):
        for mode in ("0", "1"):
            os.environ["OESS_W128_CONV3"] = mode
            for _ in range(3):
                hip.conv2d_nhwc(x, packed, None, C, 3, 3, 1, dil, dil, out=out, tile_stats=part)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                hip.conv2d_nhwc(x, packed, None, C, 3, 3, 1, dil, dil, out=out, tile_stats=part)
            e1.record(); torch.cuda.synchronize()
            t[mode].append(e0.elapsed_time(e1) / 20 * 1e3)
    fl = 2.0 * M * C * C * 9
    a, b = statistics.median(t["0"]), statistics.median(t["1"])
    print(f"{C:4d} -> {C:4d} dil {dil}: row-halo {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s)   w128 {b:7.1f} us ({fl / b / 1e6:6.0f} TF/s)   {a / b:5.2f}x", flush=True)
