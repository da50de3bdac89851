"""OpenESSLinearProbeModel (training/linear_probe_trainer.py:79): stage-2 linear probing.  Both students are built with
`if_linear_probing` (linear_probe_trainer.py:171-178, 184-190), which freezes everything but a K->K 1x1 convolution on the
logits (models/style_networks.py:113-133,169-170 -- `decoder_scale_5` is left trainable there by oversight and therefore sits
in the optimiser with no gradient; models/deeplabv3.py:162-170,186-187).  No `if_finetuning` / `frozen_backbone` arguments."""
from ._supervised import SupervisedTrainer


class OpenESSLinearProbeModel(SupervisedTrainer):
    def backend_kwargs(self):
        return {'if_linear_probing': self.settings.if_linear_probing}

    def deeplab_kwargs(self):
        return {'if_linear_probing': self.settings.if_linear_probing}
