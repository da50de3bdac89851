#!/usr/bin/env python3
"""Throughput of the stage-2/3 trainers and of OpenESSModel at the BASELINE size (BASELINE configs[2] "openess_trainer full path"
and configs[4] "linear-probe + fine-tune"): the trainers are built through train.py's own dispatch from the synthetic YAMLs with
the sizes raised to 440 x 640, B = 8, 20 sub-windows x 100 000 events, ONE batch is prepared on the device (voxelizer included,
outside the timed region) and `train_step` is timed on it between synchronize fences, like bench.py's headline.
    python tools/bench_stage2.py [--steps 20] [--json]"""
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

PIPELINE = True

CASES = [  # name, yaml, config_option, flags
    ("finetune_frame2voxel", "finetune_dsec_synthetic.yaml", "frame2voxel", dict(if_finetuning=True)),
    ("finetune_frame2recon", "finetune_dsec_synthetic.yaml", "frame2recon", dict(if_finetuning=True)),
    ("linear_probe_frame2voxel", "finetune_dsec_synthetic.yaml", "frame2voxel", dict(if_finetuning=False, if_linear_probing=True)),
    ("linear_probe_frame2recon", "finetune_dsec_synthetic.yaml", "frame2recon", dict(if_finetuning=False, if_linear_probing=True)),
    ("openess_frame2recon_contrastive", "openess_dsec_synthetic.yaml", "frame2recon", dict(if_spatial_contrastive=True)),
]


def build(yaml_name, option, flags, tmp, B=8):
    import train
    from openess_amd.config.settings import Settings
    cfg = yaml.safe_load(open(os.path.join(ROOT, "tests", "configs", yaml_name)))
    cfg['dataset']['DSEC_events'].update(shape=[440, 640], nr_events_data=20, nr_events_window=100000)
    cfg['optim'].update(batch_size_b=B, num_epochs=1)
    cfg['hardware']['num_cpu_workers'] = 0
    cfg['checkpoint']['save_checkpoint'] = False
    cfg['dir']['log'] = tmp
    cfg['clip'].update(config_option=option, superpixel_size=100)
    path = os.path.join(tmp, "stage2.yaml")
    yaml.safe_dump(cfg, open(path, "w"))
    train.seed_everything()
    s = Settings(path, generate_log=False)
    s.ckpt_dir = tmp
    s.synthetic_length = B
    for k, v in flags.items():
        setattr(s, k, v)
    trainer, loop = train.build_trainer(s)
    return trainer, s


def measure(steps=20, warm=3, only=None):
    out = {}
    for name, yml, option, flags in CASES:
        if only and name not in only:
            continue
        with tempfile.TemporaryDirectory(prefix="oess_stage2_", dir="/tmp") as tmp:
            trainer, s = build(yml, option, flags, tmp)
            for m in trainer.models_dict.values():
                m.train()
            batch = next(iter(trainer.device_batches(trainer.train_loader_sensor_b)))
            front_step = getattr(trainer, 'front_step', None) if PIPELINE else None

            def run(n):
                # BaseTrainer.trainEpoch's order: the frozen half of step i + 1 (trainers that have one) is enqueued before the
                # trainable half of step i
                prev = front_step(batch) if front_step is not None else None
                out = None
                for i in range(n):
                    nxt = front_step(batch) if (front_step is not None and i + 1 < n) else None
                    out = trainer.train_step(batch, front=prev) if prev is not None else trainer.train_step(batch)
                    prev = nxt
                return out
            run(warm)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            losses, _, total = run(steps)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            out[name] = {"value": round(steps * s.batch_size_b / dt, 2), "unit": "event-frames/s", "ms_per_step": round(dt / steps * 1e3, 3),
                         "steps": steps, "trainer": type(trainer).__name__, "loss": round(float(total), 4)}
            del trainer, batch
            torch.cuda.empty_cache()
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--only", nargs="*")
    ap.add_argument("--one-stream", action="store_true", help="A/B: OpenESSModel's students on one stream; no frozen-front pipelining in the fine-tune / linear-probe steps")
    a = ap.parse_args()
    if a.one_stream:
        from openess_amd.training.openess_trainer import OpenESSModel
        OpenESSModel.two_streams = False
        PIPELINE = False
    r = measure(a.steps, only=a.only)
    for k, v in r.items():
        print(f"{k:34s} {v['value']:8.2f} event-frames/s  {v['ms_per_step']:8.3f} ms/step  ({v['trainer']}, loss {v['loss']})", file=sys.stderr)
    print(json.dumps(r))
