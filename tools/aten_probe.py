"""Which ATen operators still launch kernels inside one training step?  A TorchDispatchMode over one step of a bench.py workload
records every non-view aten op on CUDA tensors with the Python frames (inside this repo) that issued it -- forward and the Python
side of backward (custom autograd Functions); ops issued by autograd's own C++ nodes show up without a repo frame.
    python tools/aten_probe.py [workload=frame2recon_full]"""
import collections, os, sys, traceback, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.utils._python_dispatch import TorchDispatchMode
dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
wl = bench.Workload(sys.argv[1] if len(sys.argv) > 1 else "frame2recon_full", 0, 1, dev, bench.make_inputs(0, dev))
for _ in range(3):
    wl.one_step()
torch.cuda.synchronize()
cnt = collections.Counter()
VIEWS = ("view", "permute", "expand", "slice", "select", "detach", "alias", "as_strided", "unsqueeze", "squeeze", "reshape", "t.default",
         "transpose", "empty", "_unsafe_view", "size", "stride", "is_", "numel", "sym_", "_local_scalar", "record_stream", "lift_fresh",
         "split", "unbind", "narrow", "_version", "prim.", "set_", "storage", "resize_", "new_empty", "chunk", "result_type")


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func)
        if not any(v in name for v in VIEWS):
            ts = [a for a in list(args) + list((kwargs or {}).values()) if torch.is_tensor(a)]
            if any(t.is_cuda for t in ts):
                t = max(ts, key=lambda q: q.numel())
                fr = [f"{os.path.basename(f.filename)}:{f.lineno}" for f in traceback.extract_stack()[:-1]
                      if ("/root/repo" in f.filename or "openess_amd" in f.filename) and "aten_probe" not in f.filename][-3:]
                cnt[(name, tuple(t.shape), str(t.dtype).replace("torch.", ""), " <- ".join(fr))] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    wl.one_step()
torch.cuda.synchronize()
tot = collections.Counter()
for (name, shape, dt, where), n in cnt.items():
    tot[name] += n
print("== per operator:", dict(tot.most_common()))
for (name, shape, dt, where), n in sorted(cnt.items(), key=lambda kv: (-kv[1], kv[0][0]))[:90]:
    print(f"{n:4d} {name:34s} {dt:9s} {str(shape):26s} {where}")
