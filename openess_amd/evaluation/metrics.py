"""Mirror of evaluation/metrics.py: confusion matrix (:4-23), mIoU / accuracy (:26-37), MetricsSemseg (:39-65).
The integer histogram is a HIP kernel accumulating into a device-resident K x K int64 matrix (no per-batch
.cpu() sync as in the reference); results are identical integers."""
import torch

from .. import hip


def semseg_compute_confusion(y_hat_lbl, y_lbl, num_classes, ignore_label, out=None):
    assert torch.is_tensor(y_hat_lbl) and torch.is_tensor(y_lbl), 'Inputs must be torch tensors'
    assert y_lbl.device == y_hat_lbl.device, 'Input tensors have different device placement'
    if out is None:
        out = torch.zeros(num_classes * num_classes, dtype=torch.int64, device=y_lbl.device)
    hip.confusion_accumulate(y_hat_lbl, y_lbl, num_classes, ignore_label, out)
    return out.view(num_classes, num_classes)


def semseg_accum_confusion_to_iou(confusion_accum):
    conf = confusion_accum.double()
    diag = conf.diag()
    iou_per_class = 100 * diag / (conf.sum(dim=1) + conf.sum(dim=0) - diag).clamp(min=1e-12)
    return iou_per_class.mean(), iou_per_class


def semseg_accum_confusion_to_acc(confusion_accum):
    conf = confusion_accum.double()
    diag = conf.diag()
    return 100 * diag.sum() / (conf.sum(dim=1).sum()).clamp(min=1e-12)


class MetricsSemseg:
    def __init__(self, num_classes, ignore_label, class_names):
        self.num_classes, self.ignore_label, self.class_names = num_classes, ignore_label, class_names
        self.metrics_acc = None

    def reset(self):
        self.metrics_acc = None

    def update_batch(self, y_hat_lbl, y_lbl):
        with torch.no_grad():
            if self.metrics_acc is None:
                self.metrics_acc = torch.zeros(self.num_classes ** 2, dtype=torch.int64, device=y_lbl.device)
            hip.confusion_accumulate(y_hat_lbl, y_lbl, self.num_classes, self.ignore_label, self.metrics_acc)

    def get_metrics_summary(self):
        cm = self.metrics_acc.view(self.num_classes, self.num_classes).cpu()
        iou_mean, iou_per_class = semseg_accum_confusion_to_iou(cm)
        out = {self.class_names[i]: iou for i, iou in enumerate(iou_per_class)}
        out['miou'] = iou_mean
        out['acc'] = semseg_accum_confusion_to_acc(cm)
        out['cm'] = cm
        return out
