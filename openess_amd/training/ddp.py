"""Data-parallel glue: one process per GPU, per-rank batch, ONE collective per step -- a
bucketed all-reduce (mean) of the trainable parameters' gradients -- plus a one-time weight broadcast.
Backend-agnostic (`nccl` = RCCL over xGMI on the GPU node, `gloo` in the CPU tests).

BatchNorm batch statistics, EventPreprocessor whole-batch statistics, batch-global Dice sums and InfoNCE
negatives all stay LOCAL to the rank: the reference is single-GPU with batch 8, so a replica with local
batch 8 reproduces its semantics exactly (no SyncBN, no cross-rank negatives)."""
import torch
import torch.distributed as dist


def broadcast_module_states(modules, src=0):
    for m in modules:
        for t in list(m.parameters()) + list(m.buffers()):
            dist.broadcast(t.data, src)


class GradAllReduce:
    """Flatten -> all_reduce -> unflatten, in buckets of `bucket_bytes` (fp32 grads; 29.5 MB total for the
    frame2voxel step, ~165 MB for frame2recon).  Parameters whose grad is None (never used in the forward,
    e.g. SemSegE2VID.decoder_scale_5 / DeepLabHead.pixel_feature) are skipped exactly like AdamW skips them;
    feeding zeros instead would apply weight decay and diverge from the reference."""

    def __init__(self, params, world_size=None, bucket_bytes=32 << 20):
        self.params = [p for p in params if p.requires_grad]
        self.world = world_size if world_size is not None else (dist.get_world_size() if dist.is_initialized() else 1)
        self.bucket_bytes = bucket_bytes

    def __call__(self):
        if self.world == 1:
            return
        grads = [p.grad for p in self.params if p.grad is not None]
        bucket, size = [], 0
        for g in grads:
            bucket.append(g)
            size += g.numel() * g.element_size()
            if size >= self.bucket_bytes:
                self._reduce(bucket)
                bucket, size = [], 0
        if bucket:
            self._reduce(bucket)

    def _reduce(self, grads):
        flat = torch.cat([g.reshape(-1).float() for g in grads])
        dist.all_reduce(flat)
        flat /= self.world
        off = 0
        for g in grads:
            n = g.numel()
            g.copy_(flat[off:off + n].view_as(g))
            off += n


def shard_indices(n_samples, rank, world, epoch=0, seed=1205):
    """Rank r takes samples r, r+W, ... of a seeded permutation (shuffle=True, drop_last=True equivalent,
    training/base_trainer_ov.py:166-173)."""
    g = torch.Generator().manual_seed(seed + epoch)
    perm = torch.randperm(n_samples, generator=g)
    usable = (n_samples // world) * world
    return perm[:usable][rank::world]
