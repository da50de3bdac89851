"""Where do the small copies of a pre-training step come from?  A TorchDispatchMode over one headline step records the Python
frames (inside this repo) of every aten copy / _to_copy / fill on tensors of <= 4096 elements."""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "frame2voxel_pixel_distill", 0, 1, dev, bench.make_inputs(0, dev))
for _ in range(3):
    wl.one_step()
torch.cuda.synchronize()
cnt = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if any(k in name for k in ("copy", "fill", "zero", "add", "mul", "clone", "contiguous", "cat", "stack", "index", "slice_scatter")):
            t = next((a for a in args if torch.is_tensor(a)), None)
            if t is not None and (t.numel() <= 4096 or "copy" in name or "clone" in name or "add" in name or "mul" in name):
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1]
                      if "/root/repo" in f.filename or "openess_amd" in f.filename][-3:]
                cnt[(name, str(t.device), tuple(t.shape), " <- ".join(fr))] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    wl.one_step()
torch.cuda.synchronize()
for (name, d, shape, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:40]:
    print(f"{n:4d} {name:28s} {d:7s} {str(shape):14s} {where}")
