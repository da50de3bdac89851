#!/usr/bin/env python3
"""Micro-benchmark of the K1 voxelizer at the BASELINE size (B=8 x 20 sub-windows x 100k events,
640x480 -> 8 x 100 x 440 x 640).  Prints ms per batch and algorithmic GB/s (16 B/event +
4 B/output voxel at 480 rows = 1.239 GB per batch of 8)."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402
from tests import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--raw", type=int, default=1)
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--chunks", type=int, default=1, help="split the batch into this many calls over sample groups (record workspace reused)")
    ap.add_argument("--h2d", type=int, default=0, help="1: time the pinned-host -> device copy of the raw columns with every batch")
    a = ap.parse_args()
    C, H, W, crop, nwin, n_per, B = 5, 480, 640, 40, 20, 100000, a.B
    xs, ys, ts, ps = [], [], [], []
    for b in range(B):
        x, y, t, p = synth.dsec_raw_events(nwin * n_per, H, W, seed=1205 + b)
        xs.append(x); ys.append(y); ts.append(t); ps.append(p)
    x = torch.from_numpy(np.concatenate(xs)).cuda(); y = torch.from_numpy(np.concatenate(ys)).cuda()
    t = torch.from_numpy(np.concatenate(ts)).cuda(); p = torch.from_numpy(np.concatenate(ps)).cuda()
    maps = torch.from_numpy(synth.rectify_map(H, W)[None]).cuda()
    seg_map = torch.zeros(B * nwin, dtype=torch.int32).cuda()
    so = torch.arange(0, (B * nwin + 1) * n_per, n_per, dtype=torch.int64)
    so_dev = so.cuda()
    out = torch.empty((B * nwin * C, H - crop, W), dtype=torch.float32, device="cuda")
    if not a.raw:
        rm = maps[0][y.long(), x.long()]
        xf, yf = rm[:, 0].contiguous(), rm[:, 1].contiguous()
        pf = p.float()
        tf = torch.empty_like(xf)
        for s in range(B * nwin):
            tt = (t[s * n_per:(s + 1) * n_per] - t[s * n_per]).double().float()
            tf[s * n_per:(s + 1) * n_per] = tt / tt[-1]

    if a.h2d:       # the loader's pinned buffers (DataLoader(pin_memory=True)) -> non_blocking copies -> voxelizer
        hx, hy, ht, hp = (v.cpu().pin_memory() for v in (x, y, t, p))

    def run():
        if a.raw and a.h2d:
            dx, dy, dt, dp = (v.to("cuda", non_blocking=True) for v in (hx, hy, ht, hp))
            hip.voxelize_dsec_raw(dx, dy, dt, dp, maps, seg_map, so, C, H, W, crop_rows=crop, out=out)
        elif a.raw and a.chunks > 1:
            per = B * nwin // a.chunks
            for c in range(a.chunks):
                e0_, e1_ = int(so[c * per]), int(so[(c + 1) * per])
                hip.voxelize_dsec_raw(x[e0_:e1_], y[e0_:e1_], t[e0_:e1_], p[e0_:e1_], maps, seg_map[c * per:(c + 1) * per],
                                      so[c * per:(c + 1) * per + 1] - so[c * per], C, H, W, crop_rows=crop, out=out[c * per * C:(c + 1) * per * C])
        elif a.raw:
            hip.voxelize_dsec_raw(x, y, t, p, maps, seg_map, so, C, H, W, crop_rows=crop, out=out)
        else:
            hip.voxelize_trilinear(xf, yf, pf, tf, so, C, H, W, crop_rows=crop, out=out)

    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.iters
    alg = B * (16 * nwin * n_per + 4 * nwin * C * H * W)
    print(f"voxelize B={B} raw={a.raw} h2d={a.h2d} chunks={a.chunks}: {ms:.3f} ms/batch  algorithmic {alg / 1e9:.3f} GB -> {alg / ms / 1e6:.1f} GB/s "
          f"({alg / ms / 1e6 / 8000 * 100:.1f}% of 8 TB/s)  {B / ms * 1000:.0f} event-frames/s")


if __name__ == "__main__":
    main()
