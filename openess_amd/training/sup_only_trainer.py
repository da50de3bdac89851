"""SupOnlyModel (training/sup_only_trainer.py:80): supervised-only baseline, the first branch of train.py's dispatch
(`if_supervised_only`).  Plain constructors for both students (sup_only_trainer.py:172-178, 184-189: no probing, fine-tuning or
frozen-backbone flags) and the only trainer of the three with an AMP branch: under `use_amp` the reference wraps the step in
autocast and a GradScaler (:247-252, :312-329, `scaler.update()` inside the per-optimiser loop).  This framework has ONE numeric
mode -- bf16 storage with fp32 accumulation, fp32 master weights -- so there is nothing to scale: `scaler` stays None and the
request is logged ("use_amp")."""
from ._supervised import SupervisedTrainer


class SupOnlyModel(SupervisedTrainer):
    def amp_requested(self):
        return bool(getattr(self.settings, 'use_amp', False))
