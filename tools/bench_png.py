#!/usr/bin/env python3
"""Latency of the batched GPU PNG decode at the BASELINE size: 24 maps of 440 x 640 (label, pseudo-label and
superpixel map of 8 samples) per call, for label-like content and for the worst case (noise: ~1 symbol per pixel).  Prints ms per
call next to the host path (Pillow decode + int64 widening, one thread)."""
import io
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402
from tests import png_cases  # noqa: E402


def main():
    from PIL import Image
    H, W, n = 440, 640, 24
    rng = np.random.default_rng(0)
    m = png_cases.maps(rng, H, W)
    for kind in ("labels", "blocks", "smooth", "noise"):
        files = [png_cases.pillow_bytes(np.roll(m[kind], i, axis=1)) for i in range(n)]
        blob = torch.from_numpy(np.frombuffer(b"".join(files), np.uint8).copy()).cuda()
        lens = [len(f) for f in files]
        flips = [i % 2 == 0 for i in range(n)]
        for _ in range(2):
            out, st = hip.png_decode_gray8_batch(blob, lens, H, W, flips)
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            hip.png_decode_gray8_batch(blob, lens, H, W, flips, out=out)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        t0 = time.perf_counter()
        for f in files:
            torch.from_numpy(np.array(Image.open(io.BytesIO(f)))).long()
        host = (time.perf_counter() - t0) * 1e3
        print(f"png decode {kind:7s}: {n} maps {H}x{W}, {sum(lens) / n / 1024:6.1f} KiB/file: GPU {ms:7.2f} ms/call (one wave per map), "
              f"host Pillow + int64 {host:6.1f} ms (1 thread); int64 H2D it replaces: {n * H * W * 8 / 1e6:.1f} MB")


if __name__ == "__main__":
    main()
