#!/usr/bin/env python3
"""Same-box A/B of the differentiable teacher head (BASELINE configs[2], frame2voxel_full): the fused node
(hip.bilinear_l2norm_train: one forward kernel that also leaves 1 / |x|) against the two-node path it replaced
(bilinear_resize -> l2_normalize), alternating runs of the whole step."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from openess_amd import hip

dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
inputs = bench.make_inputs(0, dev)
fused = hip.bilinear_l2norm_train
two_node = lambda x, scale=4: hip.l2_normalize(hip.bilinear_resize(x, scale_factor=scale, align_corners=True))
wl = bench.Workload("frame2voxel_full", 0, 1, dev, inputs)
for rep in range(3):
    for name, fn in (("fused", fused), ("two-node", two_node)) if os.environ.get("AB_TWO_NODE") else (("fused", fused),):
        hip.bilinear_l2norm_train = fn
        dt, loss, _ = wl.timed(20, 3)
        print(f"{name:9s} {8 * 20 / dt:7.2f} event-frames/s  {dt / 20 * 1e3:7.3f} ms/step  loss {loss:.4f}", flush=True)
