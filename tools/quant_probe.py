#!/usr/bin/env python3
"""TFLOP/s of one conv layer against its TILE COUNT (tile quantisation over the 512 / 768 workgroup slots): the image is
W = 128 pixels wide (stride 1) so that H = number of 128-row tiles; prints 128-row tile count, time, TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402

CASES = [  # name, Cin, Cout, R, stride, pad, dil
    ("3x3 256->256", 256, 256, 3, 1, 1, 1),
    ("3x3 128->128", 128, 128, 3, 1, 1, 1),
    ("5x5s2 128->256", 128, 256, 5, 2, 2, 1),
    ("5x5s2 64->128", 64, 128, 5, 2, 2, 1),
    ("1x1 1024->256", 1024, 256, 1, 1, 0, 1),
]


def main():
    for name, Cin, Cout, R, st, pad, dil in CASES:
        nt = (Cout + 127) // 128
        for tiles in (128, 256, 384, 512, 550, 640, 768, 1024, 1100, 1280, 1536, 2048, 2200, 2560, 4096):
            mt = tiles // nt
            H, W = (mt, 128) if st == 1 else (2 * mt, 256)
            x = torch.randn(1, H, W, Cin, device="cuda").bfloat16()
            w = torch.randn(Cout, Cin, R, R, device="cuda") * 0.05
            pk = hip.pack_conv_weight(w)
            Ho = (H + 2 * pad - dil * (R - 1) - 1) // st + 1
            Wo = (W + 2 * pad - dil * (R - 1) - 1) // st + 1
            out = torch.empty(1, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
            for _ in range(3):
                hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            n = 20
            e0.record()
            for _ in range(n):
                hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / n
            fl = 2.0 * Ho * Wo * Cout * Cin * R * R
            print(f"{name:16s} tiles128 {Ho * Wo // 128 * nt:5d}  {ms * 1e3:8.1f} us  {fl / ms / 1e9:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
