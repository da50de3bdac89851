#!/usr/bin/env python3
"""python train.py --settings_file <yaml>  -- same flag and the same trainer dispatch order as the
reference's train.py:26-50 (if_supervised_only -> if_pretraining -> if_finetuning -> if_linear_probing ->
OpenESSModel), seeds 1205 (train.py:15-23).  Multi-GPU: launch with torch.distributed.run."""
import argparse
import os
import random

import numpy as np
import torch
import torch.distributed as dist

from openess_amd.config.settings import Settings

SEED = 1205


def seed_everything(seed=SEED):
    np.random.seed(seed)
    random.seed(seed)
    os.environ['PYTHONHASHSEED'] = str(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def build_trainer(settings):
    from openess_amd.training.finetune_trainer import OpenESSFineTuneModel
    from openess_amd.training.linear_probe_trainer import OpenESSLinearProbeModel
    from openess_amd.training.sup_only_trainer import SupOnlyModel
    from openess_amd.training.openess_trainer import OpenESSModel
    from openess_amd.training.pretrain_trainer import OpenESSPretrainModel
    if settings.if_supervised_only:
        return SupOnlyModel(settings=settings), 'training'
    if getattr(settings, 'if_pretraining', False):
        return OpenESSPretrainModel(settings=settings), 'pretraining'
    if getattr(settings, 'if_finetuning', False):
        return OpenESSFineTuneModel(settings=settings), 'training'
    if settings.if_linear_probing:
        return OpenESSLinearProbeModel(settings=settings), 'training'
    return OpenESSModel(settings=settings), 'training'


def main():
    parser = argparse.ArgumentParser(description='Train network.')
    parser.add_argument('--settings_file', help='Path to settings yaml', required=True)
    args = parser.parse_args()
    seed_everything()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', '0')))
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl')
    settings = Settings(args.settings_file, generate_log=int(os.environ.get('RANK', '0')) == 0)
    trainer, loop = build_trainer(settings)
    getattr(trainer, loop)()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
