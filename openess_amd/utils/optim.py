"""torch.optim.AdamW with its step() on the multi-tensor HIP kernel (oess_adamw_multi_f32): same constructor, same
state layout ('step', 'exp_avg', 'exp_avg_sq'), so optimiser state_dicts interchange with the library class the
reference builds (training/pretrain_trainer.py:231-243, openess_trainer.py, finetune trainers).  Options the kernel
does not implement (amsgrad, maximize, capturable, differentiable, non-fp32 or CPU parameters) take the library path."""
import math

import numpy as np
import torch
from torch.autograd.graph import increment_version

from .. import _lib

# elements per workgroup of the multi-tensor kernel: the 7.4 M trainable parameters of the pre-training step are ~1 800 workgroups
# (65 536 made 127 workgroups on 256 CUs with 256 dependent iterations each: 165 us for 206 MB of traffic)
CHUNK = 4096


def step_key(plist):
    """Identity of a launch group inside a param group: the parameters it updates."""
    return tuple(id(p) for p in plist)


class AdamW(torch.optim.AdamW):
    def _hip_ok(self, group, params):
        return (not group.get('amsgrad', False) and not group.get('maximize', False) and not group.get('capturable', False)
                and not group.get('differentiable', False)
                and all(p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.dtype == torch.float32
                        and not p.grad.is_sparse for p in params))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        groups = [(g, [p for p in g['params'] if p.grad is not None]) for g in self.param_groups]
        if not all(self._hip_ok(g, ps) for g, ps in groups if ps):
            super().step()                                             # library path for anything unusual (all groups)
            return loss
        lib = _lib.load()
        for group, params in groups:
            if not params:
                continue
            beta1, beta2 = group['betas']
            lr = float(group['lr'])
            # identical step count for every parameter of a group that has always had gradients together; parameters
            # that join later (their grad was None before) get their own launch
            by_step = {}
            for p in params:
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                by_step.setdefault(int(st['step'].item()), []).append(p)
            for step, plist in by_step.items():
                dev = plist[0].device
                rows, cmap, keep_alive = [], [], []
                for ti, p in enumerate(plist):
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    keep_alive.append(g)       # a contiguous temporary must outlive the launch (the caching allocator would
                    #                            hand its block to the `table` / `chunks` uploads below)
                    rows.append((p.data_ptr(), g.data_ptr(), st['exp_avg'].data_ptr(), st['exp_avg_sq'].data_ptr(), p.numel()))
                    cmap.extend((ti, c) for c in range((p.numel() + CHUNK - 1) // CHUNK))
                # pointer table on the device: re-used while the pointers repeat (parameters and state never move, and the
                # caching allocator hands backward the same gradient blocks step after step); a changed table is staged through
                # PINNED memory -- an upload from pageable memory blocks the host until the stream has drained, i.e. it would
                # put a full device synchronisation at the end of every training step
                key = (id(group), step_key(plist))
                cached = self._tables.get(key) if hasattr(self, '_tables') else None
                if cached is None or cached[0] != rows:
                    if not hasattr(self, '_tables'):
                        self._tables = {}
                    table = torch.from_numpy(np.asarray(rows, dtype=np.int64)).pin_memory().to(dev, non_blocking=True)
                    chunks = torch.from_numpy(np.asarray(cmap, dtype=np.int32)).pin_memory().to(dev, non_blocking=True)
                    self._tables[key] = (rows, table, chunks)
                else:
                    _, table, chunks = cached
                bc1 = 1.0 - beta1 ** step
                bc2_sqrt = math.sqrt(1.0 - beta2 ** step)
                _lib.check(lib.oess_adamw_multi_f32(table.data_ptr(), len(rows), chunks.data_ptr(), len(cmap), CHUNK, lr,
                                                    float(beta1), float(beta2), float(group['eps']), float(group['weight_decay']),
                                                    float(bc1), float(bc2_sqrt), torch.cuda.current_stream(dev).cuda_stream),
                           "oess_adamw_multi_f32")
                # the kernel wrote through raw pointers: tell autograd (and the packed-weight caches keyed on ._version)
                increment_version(list(plist) + [self.state[p][k] for p in plist for k in ('exp_avg', 'exp_avg_sq')])
                for g in keep_alive:           # stream-ordered: blocks freed from here on are reused only after the kernel
                    g.record_stream(torch.cuda.current_stream(dev))
                del keep_alive
        return loss
