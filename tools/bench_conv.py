#!/usr/bin/env python3
"""TFLOP/s of the implicit-GEMM conv on the hot shapes of the frame2voxel step (DSEC, B=8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402

SHAPES = [
    # name, B, H, W, Cin, Cout, R, stride, pad, dil
    ("e2vid head 5x5 8->32", 8, 440, 640, 8, 32, 5, 1, 2, 1),
    ("e2vid enc0 5x5s2 32->64", 8, 440, 640, 32, 64, 5, 2, 2, 1),
    ("e2vid gates0 3x3 128->256", 8, 220, 320, 128, 256, 3, 1, 1, 1),
    ("e2vid enc1 5x5s2 64->128", 8, 220, 320, 64, 128, 5, 2, 2, 1),
    ("e2vid gates1 3x3 256->512", 8, 110, 160, 256, 512, 3, 1, 1, 1),
    ("e2vid enc2 5x5s2 128->256", 8, 110, 160, 128, 256, 5, 2, 2, 1),
    ("e2vid gates2 3x3 512->1024", 8, 55, 80, 512, 1024, 3, 1, 1, 1),
    ("r50 l1 1x1 256->64", 8, 110, 160, 256, 64, 1, 1, 0, 1),
    ("r50 l1 3x3 64->64", 8, 110, 160, 64, 64, 3, 1, 1, 1),
    ("r50 l1 1x1 64->256", 8, 110, 160, 64, 256, 1, 1, 0, 1),
    ("r50 l3 3x3d4 256->256", 8, 110, 160, 256, 256, 3, 1, 4, 4),
    ("r50 l3 1x1 256->1024", 8, 110, 160, 256, 1024, 1, 1, 0, 1),
    ("r50 l4 1x1 2048->512", 8, 110, 160, 2048, 512, 1, 1, 0, 1),
    ("r50 l4 3x3d8 512->512", 8, 110, 160, 512, 512, 3, 1, 8, 8),
    ("dec 3x3 256->256 @55x80", 8, 55, 80, 256, 256, 3, 1, 1, 1),
    ("dec 3x3 64->32 @440x640", 8, 440, 640, 64, 32, 3, 1, 1, 1),
]


def main():
    for name, B, H, W, Cin, Cout, R, st, pad, dil in SHAPES:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        w = torch.randn(Cout, Cin, R, R, device="cuda") * 0.05
        pk = hip.pack_conv_weight(w)
        Ho = (H + 2 * pad - dil * (R - 1) - 1) // st + 1
        Wo = (W + 2 * pad - dil * (R - 1) - 1) // st + 1
        out = torch.empty(B, Ho, Wo, Cout, device="cuda", dtype=torch.bfloat16)
        for _ in range(3):
            hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        n = 10
        e0.record()
        for _ in range(n):
            hip.conv2d_nhwc(x, pk, None, Cout, R, R, st, pad, dil, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / n
        fl = 2.0 * B * Ho * Wo * Cout * Cin * R * R
        xt = x.permute(0, 3, 1, 2)
        wt = w.bfloat16()
        for _ in range(2):
            torch.nn.functional.conv2d(xt, wt, stride=st, padding=pad, dilation=dil)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            torch.nn.functional.conv2d(xt, wt, stride=st, padding=pad, dilation=dil)
        e1.record()
        torch.cuda.synchronize()
        ms_t = e0.elapsed_time(e1) / n
        print(f"{name:30s} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s   | torch/MIOpen bf16 NHWC {ms_t:8.3f} ms {fl / ms_t / 1e9:8.1f} TF/s")


if __name__ == "__main__" and "--wgrad" not in sys.argv:
    main()


def bench_wgrad():
    shapes = [("dec 3x3 256->256 @55x80", 8, 55, 80, 256, 256, 3, 1, 1), ("dec 3x3 256->128 @110x160", 8, 110, 160, 256, 128, 3, 1, 1),
              ("dec 3x3 128->64 @220x320", 8, 220, 320, 128, 64, 3, 1, 1), ("dec 3x3 64->32 @440x640", 8, 440, 640, 64, 32, 3, 1, 1),
              ("head 1x1 32->256 @440x640", 8, 440, 640, 32, 256, 1, 1, 0)]
    for name, B, H, W, Cin, Cout, R, st, pad in shapes:
        x = torch.randn(B, H, W, Cin, device="cuda").bfloat16()
        gy = torch.randn(B, H, W, Cout, device="cuda").bfloat16()
        w = torch.randn(Cout, Cin, R, R, device="cuda").bfloat16()
        fl = 2.0 * B * H * W * Cout * Cin * R * R
        for tag, fn in (("hip", lambda: hip.conv2d_wgrad(x, gy, Cout, Cin, R, R, st, pad, 1)),
                        ("aten", lambda: torch.ops.aten.convolution_backward(gy.permute(0, 3, 1, 2), x.permute(0, 3, 1, 2), w, None, [st, st], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))):
            for _ in range(2):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                fn()
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 5
            print(f"wgrad {name:28s} {tag:5s} {ms:8.3f} ms {fl / ms / 1e9:8.1f} TF/s")


if __name__ == "__main__" and "--wgrad" in sys.argv:
    bench_wgrad()
