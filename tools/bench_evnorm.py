import torch, sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/openess_amd") else os.getcwd())
from openess_amd import hip
ev = torch.randn(8, 100, 440, 640, device="cuda")
ev[ev.abs() < 1.0] = 0
for _ in range(3): hip.event_slice_to_nhwc8(ev, 5, 5)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(20): hip.event_slice_to_nhwc8(ev, 5 * i, 5)
e1.record(); torch.cuda.synchronize()
print("stats+relayout per window: %.1f us" % (e0.elapsed_time(e1) / 20 * 1e3))
