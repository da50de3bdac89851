#!/usr/bin/env python3
"""8-GPU dry run of the HOST side on one GPU box: P loader pools (one per would-be rank: PinnedRingLoader with 2
workers each, batch 8 at the BASELINE size = 208 MB of raw event columns + frames / label maps per batch) are drained round-robin
by ONE process that does what BaseTrainer.device_batches does with a batch -- enqueue its host->device copies on a side stream
and hand the slot back behind their HIP event.  No training step runs: the figure is what the host (collate into the pinned
rings, queues, the single PCIe link of this box) sustains, to be read against P x the resident-input headline.
    python tools/bench_host_pools.py [--pools 8] [--workers 2] [--batches 12] [--json]
(On a real 8-GPU node each rank has its own process and its own link; here the 8 pools share one process and one link, so the
number is a lower bound for the collate side and says nothing about xGMI.)"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _bytes(o):
    if torch.is_tensor(o):
        return o.numel() * o.element_size()
    if isinstance(o, dict):
        return sum(_bytes(v) for v in o.values())
    if isinstance(o, (list, tuple)):
        return sum(_bytes(v) for v in o)
    return 0


def _to_dev(o):
    if torch.is_tensor(o):
        return o.to("cuda", non_blocking=True)
    if isinstance(o, dict):
        return {k: _to_dev(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return type(o)(_to_dev(v) for v in o)
    return o


def measure(pools=8, workers=2, batches=12, B=8, warm=2, copy=True):
    from openess_amd.datasets.ring_loader import PinnedRingLoader
    from openess_amd.datasets.synthetic_events import SyntheticEvents
    ds = SyntheticEvents(length=(batches + warm) * B, pool=16)          # shared copy-on-write by every pool's workers
    loaders = [PinnedRingLoader(ds, batch_size=B, shuffle=False, drop_last=True, num_workers=workers, slots=workers + 2) for _ in range(pools)]
    pinned = all(ld.pinned for ld in loaders)
    side = torch.cuda.Stream() if copy else None
    try:
        its = [iter(ld) for ld in loaders]
        n, nbytes, t0 = 0, 0, None
        for k in range(batches + warm):
            if k == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            for ld, it in zip(loaders, its):
                b = next(it)
                if k >= warm:
                    n += 1
                    nbytes += _bytes(b)
                if copy:
                    with torch.cuda.stream(side):
                        d = _to_dev(b)
                        ev = torch.cuda.Event()
                        ev.record(side)
                    ld.consumed_after(ev)
                    del d
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    finally:
        for ld in loaders:
            ld.close()
    return {"value": round(n * B / dt, 1), "unit": "event-frames/s", "pools": pools, "workers_per_pool": workers, "batches_per_pool": batches,
            "host_gb_per_s": round(nbytes / dt / 1e9, 2), "mb_per_batch": round(nbytes / max(n, 1) / 1e6, 1), "pinned": pinned,
            "h2d_copies": bool(copy),
            "note": "P PinnedRingLoader pools drained round-robin by one process; every batch copied to the device on a side stream out of "
                    "its pinned slot (one PCIe link for all pools on this box); no training step"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pools", type=int, default=8)
    ap.add_argument("--workers", type=int, default=2)
    ap.add_argument("--batches", type=int, default=12)
    ap.add_argument("--no-copy", action="store_true", help="drain the pools without host->device copies (collate side alone)")
    ap.add_argument("--json", action="store_true")
    a = ap.parse_args()
    r = measure(a.pools, a.workers, a.batches, copy=not a.no_copy)
    print(json.dumps(r) if a.json else r)
