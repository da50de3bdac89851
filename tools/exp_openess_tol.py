#!/usr/bin/env python3
"""Experiment: where does the InfoNCE gap of the frame2recon-style steps come from?
OpenESSModel step vs the fp32 oracle over two optimiser steps, for (a) bf16 full-resolution features (default), (b) fp32
full-resolution features (deeplabv3_resnet50.feats_fp32), and the oracle with bf16-storage emulation vs the plain oracle."""
import os, sys, copy
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import train
from openess_amd.config.settings import Settings
from openess_amd.models.deeplabv3 import deeplabv3_resnet50
from oracle.step import OracleOpenESSStep
from oracle import nets as on
from tests.synth import damp_residual, fill_by_name

CFG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "configs")


def inputs(K, H, W, seed=8):
    torch.manual_seed(seed)
    B = 2
    frame, recon = torch.rand(B, 3, H, W), torch.rand(B, 3, H, W)
    pl = torch.randint(0, K, (B, H // 4, W // 4)).repeat_interleave(4, 1).repeat_interleave(4, 2)
    sp = torch.randint(0, 45, (B, H // 8, W // 8)).repeat_interleave(8, 1).repeat_interleave(8, 2)
    return frame, recon, pl, sp


def fill(mods):
    for name in ('model_recon', 'model_frame'):
        m = mods[name]
        fill_by_name(m, 500 + len(name) + (7 if name == 'model_frame' else 0), sorted(m.state_dict().keys()))
        damp_residual(m)
        m.classifier.ASPP.project[3].p = 0.0


def run_oracle(emulate, seed, lr):
    ref = OracleOpenESSStep(11, True, lr_recon=lr, lr_frame=lr)
    fill(ref.modules())
    if emulate:
        for m in ref.modules().values():
            on.emulate_bf16_storage(m)
    frame, recon, pl, sp = inputs(11, 64, 96, seed)
    return [{k: float(v) for k, v in ref.train_step((frame, None, recon, pl, sp))[0].items()} for _ in range(2)]


def run_hip(fp32, seed, lr):
    deeplabv3_resnet50.feats_fp32 = fp32
    train.seed_everything()
    s = Settings(os.path.join(CFG, "openess_dsec_synthetic.yaml"), generate_log=False)
    s.ckpt_dir = "/tmp/exp_openess"
    s.lr_recon = s.lr_frame = lr
    trainer, _ = train.build_trainer(s)
    fill(trainer.models_dict)
    frame, recon, pl, sp = inputs(11, 64, 96, seed)
    out = [{k: float(v) for k, v in trainer.train_step((frame.cuda(), None, recon.cuda(), pl.cuda(), sp.cuda(), None))[0].items()} for _ in range(2)]
    deeplabv3_resnet50.feats_fp32 = False
    return out


torch.set_num_threads(16)
for lr in (5e-4, 1e-4, 2e-5):
  for seed in (8, 9, 10):
    o = run_oracle(False, seed, lr)
    e = run_oracle(True, seed, lr)
    a = run_hip(False, seed, lr)
    b = run_hip(True, seed, lr)
    for it in range(2):
        for k in sorted(o[it]):
            r = lambda x: (x[it][k] - o[it][k]) / abs(o[it][k]) * 100
            print(f"lr {lr:g} seed {seed} step {it} {k:22s} oracle {o[it][k]:10.4f}  emul {r(e):+7.2f} %  hip-bf16-feats {r(a):+7.2f} %  hip-fp32-feats {r(b):+7.2f} %")
