#!/usr/bin/env python3
"""Is the idle time rocprofv3 shows in front of large kernels real?  Alternates the fused ConvLSTM launch of level 1 with the
5x5 stride-2 encoder conv of level 2 (the recurrent loop's pattern) N times: wall time per pair by HIP events around the whole
loop, to be compared with the kernel durations and gaps of a `rocprofv3 --kernel-trace` run of this same script."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402


def main():
    B, H, W, C = 8, 220, 320, 64
    xh = [torch.randn(B, H, W, 2 * C, device="cuda").bfloat16() for _ in range(2)]
    wg = torch.randn(4 * C, 2 * C, 3, 3, device="cuda") * 0.03
    pg = hip.pack_conv_weight(wg, flip=2)
    cell = torch.zeros(B, H, W, C, device="cuda")
    we = torch.randn(128, 64, 5, 5, device="cuda") * 0.03
    pe = hip.pack_conv_weight(we)
    out = torch.empty(B, H // 2, W // 2, 128, device="cuda", dtype=torch.bfloat16)
    bias = torch.zeros(4 * C, device="cuda")

    def pair(i):
        hip.convlstm_fused(xh[i & 1], pg, bias, cell, xh[1 - (i & 1)][..., C:], 3, 1)
        hip.conv2d_nhwc(xh[1 - (i & 1)][..., C:], pe, None, 128, 5, 5, 2, 2, 1, relu=True, out=out)

    for i in range(4):
        pair(i)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 40
    e0.record()
    for i in range(n):
        pair(i)
    e1.record()
    torch.cuda.synchronize()
    print(f"wall per (ConvLSTM level 1 + encoder-1 conv) pair: {e0.elapsed_time(e1) / n * 1e3:.1f} us")


if __name__ == "__main__":
    main()
