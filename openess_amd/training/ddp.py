"""Data-parallel glue: one process per GPU, per-rank batch, ONE collective per step -- a
bucketed all-reduce (mean) of the trainable parameters' gradients -- plus a one-time weight broadcast.
Backend-agnostic (`nccl` = RCCL over xGMI on the GPU node, `gloo` in the CPU tests).

BatchNorm batch statistics, EventPreprocessor whole-batch statistics, batch-global Dice sums and InfoNCE
negatives all stay LOCAL to the rank: the reference is single-GPU with batch 8, so a replica with local
batch 8 reproduces its semantics exactly (no SyncBN, no cross-rank negatives).  BatchNorm RUNNING statistics are
per-rank too (rank 0's are the ones a checkpoint stores).

How the reduction is laid out for xGMI (7 point-to-point links per GPU, ring collectives are per-link bound, so few
large messages beat many small ones):
  * gradients live in pre-flattened fp32 bucket buffers (default 25 MB: 2 buckets for the 29.5 MB frame2voxel step,
    7 for the ~165 MB frame2recon step); `p.grad` of every trainable parameter is a VIEW into its bucket, so autograd
    accumulates straight into the message buffer -- no `cat`, no copy back;
  * buckets are filled in reverse registration order (the order backward produces gradients) and each bucket's
    all-reduce is launched ASYNCHRONOUSLY from a post-accumulate-grad hook the moment its last gradient lands, on the
    process group's own stream, i.e. under the rest of backward; `finish()` (== calling the reducer) launches what is
    left, waits, and gives parameters that received no gradient their `grad = None` back (AdamW must keep skipping
    them: SemSegE2VID.decoder_scale_5, DeepLabHead.pixel_feature);
  * the mean is the collective's own AVG where the backend has it (RCCL), one in-place divide otherwise (gloo).
"""
import torch
import torch.distributed as dist


def broadcast_module_states(modules, src=0):
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


class _Bucket:
    __slots__ = ("flat", "params", "views", "pending", "handle", "launched")

    def __init__(self, params, device):
        pad = lambda n: (n + 63) // 64 * 64                  # noqa: E731  every slice starts 256-byte aligned
        self.flat = torch.zeros(sum(pad(p.numel()) for p in params), dtype=torch.float32, device=device)
        self.params, self.views, off = params, [], 0
        for p in params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += pad(p.numel())
        self.pending, self.handle, self.launched = 0, None, False


class GradAllReduce:
    def __init__(self, params, world_size=None, bucket_bytes=25 << 20, overlap=True, force_buckets=False):
        """overlap=True launches a bucket's all-reduce from the autograd hook as soon as every parameter of the bucket has
        received its gradient ONCE: valid for one backward() per step (what every trainer of this path does).  With several
        backward() calls / gradient accumulation per step pass overlap=False (buckets then go at finish()); a second
        accumulation into a bucket whose collective is already in flight raises instead of racing with it.
        force_buckets=True keeps the whole bucket / hook / collective machinery at world size 1 (a 1-rank process group:
        exercises the RCCL call path on a single GPU)."""
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.active = self.world > 1 or (force_buckets and dist.is_initialized())
        seen, self.params = set(), []
        for p in params:
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                self.params.append(p)
        self.overlap = overlap
        self.buckets, self._where, self._hooks = [], {}, []
        self._avg = None
        self._armed = False
        if not self.active:
            return
        for p in self.params:
            if p.dtype != torch.float32:
                raise TypeError("GradAllReduce buckets are fp32 (master weights / gradients are fp32 on this path)")
        cur, size = [], 0
        for p in reversed(self.params):                       # backward order ~ reverse registration order
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(_Bucket(cur, cur[0].device))
                cur, size = [], 0
        if cur:
            self.buckets.append(_Bucket(cur, cur[0].device))
        for bi, b in enumerate(self.buckets):
            for pi, p in enumerate(b.params):
                self._where[id(p)] = (bi, pi)
                self._hooks.append(p.register_post_accumulate_grad_hook(self._on_grad))

    # ------------------------------------------------------------------ per step
    def prepare(self):
        """Call after `optimizer.zero_grad()`: zero the bucket buffers (one memset each) and point every trainable
        parameter's .grad at its slice, so that backward accumulates into the message buffers."""
        if not self.active:
            return
        for b in self.buckets:
            b.flat.zero_()
            b.pending, b.handle, b.launched = len(b.params), None, False
            for p, v in zip(b.params, b.views):
                p.grad = v
        self._touched = set()
        self._armed = True

    def _launch(self, b):
        b.launched = True
        if self._avg is None:
            self._avg = dist.get_backend() == "nccl"
        op = dist.ReduceOp.AVG if self._avg else dist.ReduceOp.SUM
        b.handle = dist.all_reduce(b.flat, op=op, async_op=True)

    def _on_grad(self, p):
        if not self._armed:
            return
        bi, pi = self._where[id(p)]
        b = self.buckets[bi]
        if p.grad is not b.views[pi]:                 # someone re-assigned .grad (zero_grad after prepare): fold it back in
            b.views[pi].copy_(p.grad)
            p.grad = b.views[pi]
        if id(p) in self._touched:
            if b.launched:
                # the in-place `grad +=` that just ran on the compute stream raced with the bucket's in-flight all-reduce on the
                # communicator's stream: the reduced values are undefined.  Fail loudly.
                raise RuntimeError("GradAllReduce(overlap=True): a parameter received a second gradient accumulation after its "
                                   "bucket's all-reduce was launched (several backward() calls / gradient accumulation in one step); "
                                   "construct the reducer with overlap=False for that pattern")
            return
        self._touched.add(id(p))
        b.pending -= 1
        if b.pending == 0 and self.overlap and not b.launched:
            self._launch(b)

    def finish(self):
        if not self.active:
            return
        if not self._armed:                           # prepare() was not called this step: gather whatever .grad holds now
            self._gather_unprepared()
        for b in self.buckets:                        # buckets holding never-used parameters (or overlap off) go now, in order
            if not b.launched:
                self._launch(b)
        for b in self.buckets:
            b.handle.wait()
            if not self._avg:
                b.flat.div_(self.world)
        for b in self.buckets:
            for p in b.params:
                if id(p) not in self._touched:
                    p.grad = None                     # no gradient this step on ANY rank (same graph everywhere): AdamW skips it
        self._armed = False

    __call__ = finish

    def _gather_unprepared(self):
        self._touched = set()
        for b in self.buckets:
            b.pending, b.handle, b.launched = 0, None, False
            for p, v in zip(b.params, b.views):
                if p.grad is None:
                    v.zero_()
                    continue
                if p.grad is not v:
                    v.copy_(p.grad)
                    p.grad = v
                self._touched.add(id(p))

    def close(self):
        """Detach from the parameters (hooks removed, .grad views released): needed before a second reducer takes the same
        parameters."""
        for h in self._hooks:
            h.remove()
        self._hooks, self._armed = [], False

    def exposed_bytes(self):
        return sum(b.flat.numel() * 4 for b in self.buckets)


def shard_indices(n_samples, rank, world, epoch=0, seed=1205):
    """Rank r takes samples r, r+W, ... of a seeded permutation (shuffle=True, drop_last=True equivalent,
    training/base_trainer_ov.py:166-173)."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n_samples, generator=g)
    usable = (n_samples // world) * world
    return perm[:usable][rank::world]
