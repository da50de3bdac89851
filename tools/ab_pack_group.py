#!/usr/bin/env python3
"""Same-box A/B of the grouped weight repacking (engine.PackedWeight.refresh_stale: one launch for every stale conv operand of
the step) against one pack launch per operand, alternating runs of a whole step: python tools/ab_pack_group.py [workload]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openess_amd import engine

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
name = sys.argv[1] if len(sys.argv) > 1 else "frame2recon_full"
wl = bench.Workload(name, 0, 1, dev, bench.make_inputs(0, dev))
for rep in range(3):
    for tag, on in (("grouped", True), ("one-by-one", False)):
        engine.PackedWeight.group_enabled = on
        dt, loss, _ = wl.timed(20, 3)
        print(f"{name} {tag:10s} {8 * 20 / dt:7.2f} event-frames/s  {dt / 20 * 1e3:7.3f} ms/step", flush=True)
