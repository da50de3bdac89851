"""A/B of the grouped fused-ConvLSTM launch (the dominant kernel of the headline step) at BASELINE size: the three levels of
E2VID's recurrent encoder (B = 8; 220x320 C=64, 110x160 C=128, 55x80 C=256; Gates = 3x3 conv over cat(x, h)), one launch.

    python tools/bench_lstm_group.py [--modes 1,3] [--iters 20] [--rounds 3] [--zero]

Modes: 4 = oess_convlstm_w128_group_bf16 (256 x 256 tiles / 128 x 128 wave tiles, persistent, w128-tiled cell state); 1 / 0 =
oess_convlstm_fused_group_bf16 on 256 x 128 / 128 x 128 tiles (OESS_LSTM256, read once per process: one of them per run).  Prints TFLOP/s of the gate convolutions per mode (interleaved rounds, median and best) and the maximum
difference of hidden / cell outputs against mode 1."""
import argparse
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openess_amd import hip  # noqa: E402


def problems(B, zero, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    out = []
    for (H, W, C) in ((220, 320, 64), (110, 160, 128), (55, 80, 256)):
        xh = torch.zeros if zero else (lambda *s: torch.randn(*s, generator=g))
        buf = xh(B, H, W, 2 * C).to(torch.bfloat16).cuda()
        wt = (torch.randn(4 * C, 2 * C, 3, 3, generator=g) * (0.0 if zero else (2.0 / (18 * C)) ** 0.5)).cuda()
        bias = (torch.randn(4 * C, generator=g) * 0.1).cuda()
        packed = hip.pack_conv_weight(wt, flip=2)
        cell0 = (torch.randn(B, H, W, C, generator=g) * 0.5).cuda()
        out.append(dict(xh=buf, packed=packed, bias=bias, cell0=cell0, C=C, H=H, W=W))
    return out


def launch(mode, args):
    """mode 4 = oess_convlstm_w128_group_bf16 (w128-tiled cells), else oess_convlstm_fused_group_bf16 with OESS_LSTM256 = mode"""
    if mode == 4:
        if not hip.convlstm_w128_group(args):
            raise RuntimeError("w128 kernel did not take the problems")
    else:
        hip.convlstm_fused_group(args)


def make_args(ps, mode):
    os.environ["OESS_LSTM256"] = str(mode)
    hs = [torch.empty(p["xh"].shape[0], p["H"], p["W"], p["C"], dtype=torch.bfloat16, device="cuda") for p in ps]
    if mode == 4:
        cells = [hip.convlstm_w128_cell_relayout(p["cell0"].reshape(-1, p["C"]), p["cell0"].numel() // p["C"], p["C"], True) for p in ps]
    else:
        cells = [p["cell0"].clone() for p in ps]
    return cells, hs, [(p["xh"], p["packed"], p["bias"], c, h, 3, 1, False) for p, c, h in zip(ps, cells, hs)]


def run(ps, mode):
    cells, hs, args = make_args(ps, mode)
    launch(mode, args)
    if mode == 4:
        cells = [hip.convlstm_w128_cell_relayout(c, p["cell0"].numel() // p["C"], p["C"], False).reshape(p["cell0"].shape) for p, c in zip(ps, cells)]
    return cells, hs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--modes", default="1,4")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--zero", action="store_true")
    ap.add_argument("--stamps", action="store_true", help="print the s_memtime stamps of a W128_ABL & 8192 build (first tile of each level)")
    a = ap.parse_args()
    modes = [int(m) for m in a.modes.split(",")]
    ps = problems(a.batch, a.zero)
    flops = sum(2.0 * p["xh"].shape[0] * p["H"] * p["W"] * 4 * p["C"] * 2 * p["C"] * 9 for p in ps)
    ref_c, ref_h = run(ps, 1)
    torch.cuda.synchronize()
    for m in modes:
        c, h = run(ps, m)
        torch.cuda.synchronize()
        dc = max(float((x - y).abs().max()) for x, y in zip(c, ref_c))
        dh = max(float((x.float() - y.float()).abs().max()) for x, y in zip(h, ref_h))
        bad = sum(int((~torch.isfinite(x)).sum()) for x in c)
        print(f"mode {m}: max|cell - mode1| = {dc:.3e}  max|hidden - mode1| = {dh:.3e}  non-finite cells {bad}", flush=True)
    if a.stamps:                         # a W128_ABL & 8192 build prints its stamps on stderr after every launch
        return
    times = {m: [] for m in modes}
    for _ in range(a.rounds):
        for m in modes:
            cells, hs, args = make_args(ps, m)
            for _ in range(3):
                launch(m, args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.iters):
                launch(m, args)
            e1.record()
            torch.cuda.synchronize()
            times[m].append(e0.elapsed_time(e1) / a.iters)
    for m in modes:
        med, best = statistics.median(times[m]), min(times[m])
        print(f"mode {m}: {med * 1e3:8.1f} us median ({flops / med / 1e9:7.1f} TFLOP/s, {flops / med / 1e9 / 2500:.3f} of 2.5 PF)   "
              f"best {best * 1e3:8.1f} us ({flops / best / 1e9:7.1f} TFLOP/s)", flush=True)


if __name__ == "__main__":
    main()
