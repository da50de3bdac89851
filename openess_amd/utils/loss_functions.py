"""Mirror of utils/loss_functions.py: TaskLoss (:6-24), DiceLoss (:96-135), BinaryDiceLoss (:63-90),
NCELoss (:138-154).  Dice + cross-entropy run as ONE fused HIP pass over the logits (forward) and one
pass for the gradient; no one-hot tensor is materialised."""
import torch

from .. import hip


class TaskLoss(torch.nn.Module):
    def __init__(self, losses=['cross_entropy'], gamma=2.0, num_classes=13, alpha=None, weight=None, ignore_index=None,
                 reduction='mean'):
        super().__init__()
        if weight is not None:
            raise NotImplementedError("class weights are never passed by the reference trainers")
        self.losses = losses
        self.num_classes = num_classes
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, predict, target):
        total, _ = hip.task_loss(predict, target, self.num_classes, self.ignore_index, tuple(self.losses))
        return total


class DiceLoss(torch.nn.Module):
    def __init__(self, weight=None, num_classes=13, ignore_index=None, **kwargs):
        super().__init__()
        if weight is not None or kwargs:
            raise NotImplementedError("only the defaults (smooth=1, p=2, no weights) are used by the reference")
        self.num_classes = num_classes
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, predict, target):
        total, _ = hip.task_loss(predict, target, self.num_classes, self.ignore_index, ("dice",))
        return total


class NCELoss(torch.nn.Module):
    """PointInfoNCE: CE(k q^T / T, arange) over the S <= 100*B superpixel rows (utils/loss_functions.py:140-154)."""

    def __init__(self, temperature):
        super().__init__()
        self.temperature = temperature

    def forward(self, k, q):
        return hip.nce_loss(k, q, self.temperature)                # oess_nce_loss_fwd / _bwd
