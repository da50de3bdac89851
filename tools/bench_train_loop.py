#!/usr/bin/env python3
"""Throughput of the PRODUCT training loop (train.py's dispatch -> OpenESSPretrainModel.trainEpoch) at the BASELINE size:
DataLoader worker processes -> collate (raw event columns, 13 B/event) -> pin thread -> H2D copies + batched HIP voxelizer on
the side stream (BaseTrainer.device_batches) -> train_step.  This is the loop SURVEY 8e names as the weak-scaling limiter;
bench.py reports its rate beside the headline (`train_loop`), never as `value`.
    python tools/bench_train_loop.py [--batches 24] [--workers 10] [--no-prefetch] [--json]
The synthetic dataset serves a pool of 16 pre-generated event-frames (workers copy them like a memory-mapped recording).
A batch is 208 MB of raw columns that a worker copies, collates and hands over through shared memory: 6 workers deliver one batch
per ~60 ms, 10 or 16 one per ~47 ms = the step time (measured: 134.6 / 168.6 / 168.6 event-frames/s), hence the default of 10."""
import argparse
import json
import os
import sys
import tempfile
import time

import torch
import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(batches, workers, prefetch, B=8, contrastive=False, tmp=None, ring=True):
    import train
    from openess_amd.config.settings import Settings
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "configs", "pretrain_dsec_synthetic.yaml")))
    d = cfg['dataset']['DSEC_events']
    d.update(shape=[440, 640], nr_events_data=20, nr_events_window=100000)
    cfg['optim'].update(batch_size_b=B, num_epochs=1)
    cfg['hardware']['num_cpu_workers'] = workers
    cfg['clip'].update(if_spatial_contrastive=contrastive, superpixel_size=100)
    cfg['checkpoint']['save_checkpoint'] = False
    cfg['dir']['log'] = tmp
    path = os.path.join(tmp, "train_loop.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    train.seed_everything()
    s = Settings(path, generate_log=False)
    s.ckpt_dir = tmp
    s.synthetic_length = batches * B
    s.synthetic_pool = 16
    s.ingest_prefetch = prefetch
    s.ring_loader = bool(ring)
    trainer, loop = train.build_trainer(s)
    assert loop == 'pretraining'
    return trainer, s


def measure(batches=24, workers=4, prefetch=True, warm=4, ring=True, pipeline=True):
    with tempfile.TemporaryDirectory(prefix="oess_loop_", dir="/tmp") as tmp:
        trainer, s = build(batches + warm, workers, prefetch, tmp=tmp, ring=ring)
        for m in trainer.models_dict.values():
            m.train()
        B = s.batch_size_b
        t0, done = None, 0

        def step(batch, fr):                       # one finished step; the clock starts after `warm` of them
            nonlocal t0, done
            trainer.train_step(batch, front=fr) if fr is not None else trainer.train_step(batch)
            done += 1
            if done == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
        prev = None
        for batch in trainer.device_batches(trainer.train_loader_sensor_b):
            if pipeline:                           # BaseTrainer.trainEpoch's order: front(i + 1) enqueued before the back half of i
                fr = trainer.front_step(batch)
                if prev is not None:
                    step(*prev)
                prev = (batch, fr)
            else:
                step(batch, None)
        if prev is not None:
            step(*prev)
        n = done - warm
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        loader = trainer.train_loader_sensor_b
        kind = "PinnedRingLoader (workers collate into a pinned shared ring: one host copy)" if hasattr(loader, "consumed_after") else \
            "torch DataLoader (collate + shared-memory hand-over + pin thread: three host copies)"
        if hasattr(loader, "close"):
            loader.close()
        return {"value": round(n * B / dt, 2), "unit": "event-frames/s", "ms_per_step": round(dt / n * 1e3, 3), "steps": n,
                "loader_workers": workers, "prefetch": bool(prefetch), "loader": kind, "pipelined_steps": bool(pipeline),
                "note": "train.py's own loop: DataLoader workers -> collate (13 B/event raw columns, 208 MB/batch) -> pin thread -> "
                        "side-stream H2D + voxelizer (BaseTrainer.device_batches) -> OpenESSPretrainModel.train_step; 16-sample pool"}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=24)
    ap.add_argument("--workers", type=int, default=4)
    ap.add_argument("--dataloader", action="store_true", help="torch's DataLoader instead of the pinned ring loader (A/B)")
    ap.add_argument("--no-pipeline", action="store_true", help="one step after the other (no front(i+1) ahead of the back half of step i)")
    ap.add_argument("--no-prefetch", action="store_true")
    a = ap.parse_args()
    r = measure(a.batches, a.workers, not a.no_prefetch, ring=not a.dataloader, pipeline=not a.no_pipeline)
    print(json.dumps(r))
