"""Batch loader for the raw-event ingest path: worker processes collate STRAIGHT INTO a ring of pinned, shared host slots.

`torch.utils.data.DataLoader(..., collate_fn=collate, pin_memory=True)` moves every byte of a batch three times on the host before
the host->device copy can start: the worker's `torch.cat` of the samples' columns, the hand-over through shared memory (the
worker's result is copied into a shared segment), and the pin thread's copy into page-locked memory.  At the BASELINE size a batch is
208 MB of raw event columns plus ~110 MB of frames / label maps, so one rank needs ~10 workers to keep one MI355X busy and an
8-GPU node ~80.  Here the ring is allocated ONCE as an anonymous shared mapping, page-locked with
hipHostRegister, and inherited by the forked workers; a worker writes each sample's tensors directly into its slot of the ring
(`collate(samples, arena=...)`: the same layout code as the DataLoader path, with `out=` views of the slot) and sends back only a
small description of where everything lies.  The trainer's side-stream copies read the slot in place (it is pinned: the copies are
truly asynchronous), and the slot returns to the workers when the HIP event recorded after those copies has completed.
One host copy per byte instead of three; the reference's loader is the DataLoader of training/base_trainer_ov.py:166-173.

Same iteration contract as the DataLoader it replaces: `len()`, shuffling with a fresh permutation per epoch drawn from torch's
global generator, `drop_last`, batches delivered in order, worker seeds = base seed + worker id.  CPU-only processes (tests) get the
same loader without the page-locking."""
import mmap
import multiprocessing as mp
import os
import queue
import random
import traceback

import numpy as np
import torch

from .synthetic_events import collate

_ALIGN = 256


class Arena:
    """Bump allocator over one ring slot (a uint8 tensor); `cat` / `stack` / `put` are torch.cat / torch.stack / a copy whose result
    lives in the slot."""

    def __init__(self, buf):
        self.buf, self.off = buf, 0

    def _take(self, dtype, shape):
        n = int(np.prod(shape)) if len(shape) else 1
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        start = (self.off + _ALIGN - 1) // _ALIGN * _ALIGN
        if start + nbytes > self.buf.numel():
            raise MemoryError(f"ring slot of {self.buf.numel()} bytes is too small for this batch (needs more than {start + nbytes}); "
                              "raise PinnedRingLoader(slot_bytes=...)")
        self.off = start + nbytes
        return self.buf[start:start + nbytes].view(dtype).view(tuple(shape)), start

    def cat(self, tensors):
        shape = (sum(int(t.shape[0]) for t in tensors), *tensors[0].shape[1:])
        out, start = self._take(tensors[0].dtype, shape)
        torch.cat(tensors, out=out)
        return _InArena(out, start)

    def stack(self, tensors):
        out, start = self._take(tensors[0].dtype, (len(tensors), *tensors[0].shape))
        torch.stack(tensors, out=out)
        return _InArena(out, start)

    def put(self, t):
        out, start = self._take(t.dtype, tuple(t.shape))
        out.copy_(t)
        return _InArena(out, start)


class _InArena:
    """A tensor that lives in the ring slot: crosses the process boundary as (offset, dtype, shape), not as data."""

    def __init__(self, t, start):
        self.t, self.start = t, start


def _encode(obj):
    if isinstance(obj, _InArena):
        return ('__ring__', obj.start, str(obj.t.dtype).replace('torch.', ''), tuple(obj.t.shape))
    if isinstance(obj, dict):
        return {k: _encode(v) for k, v in obj.items()}
    if isinstance(obj, tuple):
        return ('__tuple__', [_encode(v) for v in obj])
    if isinstance(obj, list):
        return [_encode(v) for v in obj]
    return obj                      # small tensors (offsets, counts), python scalars, strings: pickled as they are


def _decode(obj, slot):
    if isinstance(obj, tuple) and len(obj) == 4 and obj[0] == '__ring__':
        _, start, dt, shape = obj
        dtype = getattr(torch, dt)
        n = int(np.prod(shape)) if len(shape) else 1
        return slot[start:start + n * torch.empty((), dtype=dtype).element_size()].view(dtype).view(shape)
    if isinstance(obj, tuple) and len(obj) == 2 and obj[0] == '__tuple__':
        return tuple(_decode(v, slot) for v in obj[1])
    if isinstance(obj, dict):
        return {k: _decode(v, slot) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_decode(v, slot) for v in obj]
    return obj


def _worker(wid, dataset, ring, slot_bytes, tasks, done, base_seed):
    torch.set_num_threads(1)
    seed = (base_seed + wid) % (1 << 63)
    random.seed(seed)
    np.random.seed(seed % (1 << 32))
    torch.manual_seed(seed)
    while True:
        job = tasks.get()
        if job is None:
            return
        epoch, seq, slot, indices = job
        try:
            samples = [dataset[i] for i in indices]
            try:
                batch = collate(samples, arena=Arena(ring[slot * slot_bytes:(slot + 1) * slot_bytes]))
                done.put((epoch, seq, slot, _encode(batch), None))
            except MemoryError:
                # this batch is larger than the slot (slot_bytes is an estimate from the first batch): collate it on the heap and
                # ship it by value -- three host copies instead of one for THIS batch, but the epoch goes on.  ('__heap__' tells
                # the consumer that the slot holds nothing.)
                done.put((epoch, seq, slot, ('__heap__', collate(samples)), None))
        except Exception:                                        # reported to the consumer, which raises
            done.put((epoch, seq, slot, None, traceback.format_exc()))


class PinnedRingLoader:
    def __init__(self, dataset, batch_size, shuffle=False, drop_last=False, num_workers=4, slot_bytes=None, slots=None, pin=None):
        if num_workers < 1:
            raise ValueError("PinnedRingLoader needs at least one worker process")
        self.dataset, self.batch_size, self.shuffle, self.drop_last = dataset, int(batch_size), bool(shuffle), bool(drop_last)
        self.num_workers = int(num_workers)
        self.slots = int(slots or (self.num_workers + 3))         # one per worker in flight, two with the consumer, one ready
        self.slot_bytes = int(slot_bytes or self._estimate_slot_bytes())
        self.slot_bytes = (self.slot_bytes + 4095) // 4096 * 4096
        self._mm = mmap.mmap(-1, self.slots * self.slot_bytes)          # anonymous MAP_SHARED: inherited by the forked workers
        self.ring = torch.frombuffer(self._mm, dtype=torch.uint8)
        self.pinned = False
        if pin is None:
            pin = torch.cuda.is_available()
        if pin:
            # page-lock the ring so that the trainer's copies from it are asynchronous DMA (hipHostRegister)
            rc = torch.cuda.cudart().cudaHostRegister(self.ring.data_ptr(), self.ring.numel(), 0)
            if int(rc) == 0:
                self.pinned = True
            else:
                # e.g. a memlock limit below the ring size: the loader still saves two of the three host copies, but the trainer's
                # host->device copies are then staged by the runtime (synchronous for large tensors) -- say so, do not die
                import warnings
                warnings.warn(f"PinnedRingLoader: hipHostRegister of the {self.ring.numel() >> 20} MB ring failed ({rc}); continuing with "
                              "a pageable ring (raise `ulimit -l` for asynchronous copies)")
        ctx = mp.get_context('fork')
        self._tasks, self._done = ctx.Queue(), ctx.Queue()
        base_seed = int(torch.empty((), dtype=torch.int64).random_().item())
        self._procs = [ctx.Process(target=_worker, args=(w, dataset, self.ring, self.slot_bytes, self._tasks, self._done, base_seed),
                                   daemon=True) for w in range(self.num_workers)]
        for p in self._procs:
            p.start()
        self._free = list(range(self.slots))
        self._busy = []                         # (slot, event | None): handed to the consumer, copies possibly still in flight
        self._last_slot = None
        self._closed = False
        self._epoch = 0                         # an iteration abandoned half-way leaves tasks in flight: their results are dropped
        self._outstanding = 0                   # tasks sent - results received, over all epochs
        self._ready = {}                        # seq -> (slot, spec) of the running iteration: received, not yet delivered
        self._heap_warned = False

    def _estimate_slot_bytes(self):
        """One batch through the plain collate on this process: its tensor bytes + 25 % (ragged event counts) + alignment slack."""
        n = min(self.batch_size, len(self.dataset))
        # the probe batch runs the dataset's (augmenting) __getitem__ in THIS process: put the random streams back afterwards
        rng = (random.getstate(), np.random.get_state(), torch.get_rng_state())
        try:
            batch = collate([self.dataset[i % len(self.dataset)] for i in range(n)])
        finally:
            random.setstate(rng[0]); np.random.set_state(rng[1]); torch.set_rng_state(rng[2])
        total = [0]

        def walk(o):
            if torch.is_tensor(o):
                total[0] += o.numel() * o.element_size() + _ALIGN
            elif isinstance(o, dict):
                for v in o.values():
                    walk(v)
            elif isinstance(o, (list, tuple)):
                for v in o:
                    walk(v)
        walk(batch)
        return int(total[0] * self.batch_size / max(n, 1) * 1.25) + (1 << 20)

    def __len__(self):
        n = len(self.dataset)
        return n // self.batch_size if self.drop_last else (n + self.batch_size - 1) // self.batch_size

    # ---- slot life cycle
    def consumed_after(self, event):
        """The consumer's copies out of the slot of the batch it received LAST were enqueued before `event` (a recorded
        torch.cuda.Event): the slot returns to the workers once the event has completed.  Without this call the slot is assumed
        free as soon as the next batch is requested (a consumer that copies synchronously)."""
        if self._last_slot is not None:
            self._busy.append((self._last_slot, event))
            self._last_slot = None

    def _reclaim(self, block):
        if self._last_slot is not None:                          # consumer did not hand over an event: done with it by now
            self._free.append(self._last_slot)
            self._last_slot = None
        still = []
        for slot, ev in self._busy:
            if ev is None or ev.query():
                self._free.append(slot)
            else:
                still.append((slot, ev))
        self._busy = still
        if block and not self._free and self._busy:
            slot, ev = self._busy.pop(0)
            ev.synchronize()
            self._free.append(slot)

    def __iter__(self):
        if self._closed:
            raise RuntimeError("PinnedRingLoader is closed")
        n = len(self.dataset)
        order = torch.randperm(n).tolist() if self.shuffle else list(range(n))
        batches = [order[i:i + self.batch_size] for i in range(0, n, self.batch_size)]
        if self.drop_last and batches and len(batches[-1]) < self.batch_size:
            batches.pop()
        self._epoch += 1
        epoch = self._epoch
        self._release_ready()                                    # slots parked by an iteration the consumer abandoned
        nxt, want, ready = 0, 0, self._ready
        try:
            while want < len(batches):
                self._reclaim(block=False)
                while nxt < len(batches) and self._free:
                    self._tasks.put((epoch, nxt, self._free.pop(), batches[nxt]))
                    nxt += 1
                    self._outstanding += 1
                if want in ready:
                    slot, spec = ready.pop(want)
                    want += 1
                    if isinstance(spec, tuple) and len(spec) == 2 and spec[0] == '__heap__':
                        self._free.append(slot)                  # the batch did not fit the slot: it came by value
                        if not self._heap_warned:
                            self._heap_warned = True
                            import warnings
                            warnings.warn(f"PinnedRingLoader: a batch did not fit its {self.slot_bytes >> 20} MB ring slot and was shipped "
                                          "through the queue (pageable, three host copies); pass slot_bytes= for the largest batch")
                        yield spec[1]
                        continue
                    self._last_slot = slot
                    yield _decode(spec, self.ring[slot * self.slot_bytes:(slot + 1) * self.slot_bytes])
                    continue
                if self._outstanding == 0:
                    if not self._free and not self._busy and self._last_slot is None:
                        raise RuntimeError("PinnedRingLoader: no ring slot can become free (all slots lost); this is a bug in the slot accounting")
                    self._reclaim(block=True)                    # every slot is with the consumer: wait for its oldest copies
                    continue
                try:
                    ep, seq, slot, spec, err = self._done.get(timeout=120)
                except queue.Empty:
                    dead = [p.pid for p in self._procs if not p.is_alive()]
                    raise RuntimeError(f"PinnedRingLoader: no batch for 120 s (dead workers: {dead})")
                self._outstanding -= 1
                if ep != epoch:                                  # left over from an iteration the consumer abandoned
                    self._free.append(slot)
                    continue
                if err is not None:
                    self._free.append(slot)
                    raise RuntimeError("PinnedRingLoader worker failed:\n" + err)
                ready[seq] = (slot, spec)
        finally:
            # normal end, `break`, an exception in the consumer, or the generator being dropped: batches received but not delivered
            # give their slots back (results still in flight are dropped, and their slots freed, by the next iteration's epoch check)
            self._release_ready()
        self._reclaim(block=False)

    def _release_ready(self):
        for slot, _ in self._ready.values():
            self._free.append(slot)
        self._ready.clear()

    def close(self):
        if self._closed:
            return
        self._closed = True
        for _ in self._procs:
            self._tasks.put(None)
        for p in self._procs:
            p.join(timeout=5)
            if p.is_alive():
                p.terminate()
        if self.pinned:
            try:
                torch.cuda.synchronize()
                torch.cuda.cudart().cudaHostUnregister(self.ring.data_ptr())
            except Exception:
                pass
            self.pinned = False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
