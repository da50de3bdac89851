"""OpenESSFineTuneModel (training/finetune_trainer.py:81): stage-2 fine-tuning of a pre-trained student on ground-truth labels.
Differences from its two siblings (reference lines): DeepLabv3 is built with `if_finetuning` and `frozen_backbone`
(finetune_trainer.py:186-193: a frozen backbone leaves only the ASPP head trainable); SemSegE2VID gets no probing flag
(:173-179); the SAM-feature slot batch[5] is moved to the device under `if_sam_distillation` but never used (:362-363) -- not
read here."""
from ._supervised import SupervisedTrainer


class OpenESSFineTuneModel(SupervisedTrainer):
    def deeplab_kwargs(self):
        s = self.settings
        return {'if_finetuning': getattr(s, 'if_finetuning', False), 'frozen_backbone': s.frozen_backbone}
