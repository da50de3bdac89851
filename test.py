#!/usr/bin/env python3
"""python test.py --settings_file <yaml>  -- evaluation entry (reference test.py:36-49): builds the fine-tune /
linear-probe trainer and runs valEpochs() only."""
import argparse

from openess_amd.config.settings import Settings
from train import seed_everything


def main():
    parser = argparse.ArgumentParser(description='Evaluate network.')
    parser.add_argument('--settings_file', help='Path to settings yaml', required=True)
    args = parser.parse_args()
    seed_everything()
    settings = Settings(args.settings_file, generate_log=True)
    from openess_amd.training.finetune_trainer import OpenESSFineTuneModel
    from openess_amd.training.linear_probe_trainer import OpenESSLinearProbeModel
    trainer = OpenESSLinearProbeModel(settings) if settings.if_linear_probing else OpenESSFineTuneModel(settings)
    metrics = trainer.valEpochs()
    print({k: float(v) for k, v in metrics.items() if k != 'cm'})


if __name__ == "__main__":
    main()
