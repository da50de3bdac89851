"""Oracle (test infrastructure): losses, superpixel pooling and metrics -- plain PyTorch fp32
on CPU (autograd supplies reference gradients) and NumPy for the integer confusion matrix.
"""
import numpy as np
import torch
import torch.nn.functional as F


# a15  DiceLoss / BinaryDiceLoss / make_one_hot   utils/loss_functions.py:43-57, 80-90, 114-135
def dice_loss(predict, target, num_classes, ignore_index=255, smooth=1.0, p=2):
    """softmax; mask = target != ignore; per class: 1 - (2*sum(p*y)+smooth)/(sum(p^p + y^p)+smooth)
    with sums over the WHOLE batch (BinaryDiceLoss flattens but then torch.sum()s everything,
    loss_functions.py:84-88); classes equal to ignore_index are skipped (:128) yet the mean still
    divides by the number of classes (:135)."""
    mask = target != ignore_index
    tgt = target * mask
    one_hot = torch.zeros(predict.shape, dtype=predict.dtype)
    one_hot.scatter_(1, tgt.unsqueeze(1), 1)
    one_hot = one_hot * mask.unsqueeze(1)
    prob = F.softmax(predict, dim=1) * mask.unsqueeze(1)
    total = 0
    for i in range(num_classes):
        if i != ignore_index:
            num = torch.sum(prob[:, i] * one_hot[:, i]) * 2 + smooth
            den = torch.sum(prob[:, i].pow(p) + one_hot[:, i].pow(p)) + smooth
            total = total + (1 - num / den)
    return total / num_classes


# a15  TaskLoss.forward   utils/loss_functions.py:17-24
def task_loss(predict, target, num_classes, ignore_index=255, losses=("dice", "cross_entropy")):
    total = 0
    if "dice" in losses:
        total = total + dice_loss(predict, target, num_classes, ignore_index)
    if "cross_entropy" in losses:
        total = total + F.cross_entropy(predict, target, ignore_index=ignore_index)
    return total


# a14  NCELoss.forward   utils/loss_functions.py:147-154
def nce_loss(k, q, temperature=0.07):
    logits = torch.mm(k, q.transpose(1, 0)) / temperature
    target = torch.arange(k.shape[0]).long()
    return F.cross_entropy(logits, target)


# a13  inline superpixel pooling   training/pretrain_trainer.py:445-465
def superpixel_pool(feat, superpixels, superpixel_size):
    """ids += b*superpixel_size (ids may exceed superpixel_size -> cross-sample collisions are
    reproduced); S = max id + 1 (sparse_coo_tensor infers the size); k = onehot @ feat_pixels;
    k /= (count + 1e-6).  feat: B x C x H x W, superpixels: B x H x W int64 -> S x C."""
    B = feat.shape[0]
    ids = torch.arange(0, B * superpixel_size, superpixel_size)[:, None, None] + superpixels
    ids = ids.flatten()
    S = int(ids.max().item()) + 1
    pix = feat.permute(0, 2, 3, 1).flatten(0, 2)
    k = torch.zeros(S, feat.shape[1], dtype=feat.dtype).index_add(0, ids, pix)
    cnt = torch.zeros(S, dtype=feat.dtype).index_add(0, ids, torch.ones(ids.shape[0], dtype=feat.dtype))
    return k / (cnt[:, None] + 1e-6)


# a16  consistency losses   training/openess_trainer.py:497-503
def consistency_losses(feat_a, feat_b, logits_a, logits_b):
    l_feat = F.l1_loss(feat_a, feat_b)
    l_pred = torch.mean(1 - F.cosine_similarity(logits_a, logits_b, dim=1))
    return l_feat, l_pred


# a17  semseg_compute_confusion / MetricsSemseg   evaluation/metrics.py:4-31, 39-65
def confusion_matrix(pred, label, num_classes, ignore_label=255):
    """bincount(pred + K*gt) over gt != ignore; rows = gt, cols = pred; int64, exact."""
    pred = np.asarray(pred).reshape(-1).astype(np.int64)
    label = np.asarray(label).reshape(-1).astype(np.int64)
    m = label != ignore_label
    x = pred[m] + num_classes * label[m]
    return np.bincount(x, minlength=num_classes ** 2).reshape(num_classes, num_classes).astype(np.int64)


def miou_acc(conf):
    conf = np.asarray(conf, dtype=np.float64)
    diag = np.diag(conf)
    iou = 100 * diag / np.clip(conf.sum(1) + conf.sum(0) - diag, 1e-12, None)
    acc = 100 * diag.sum() / max(conf.sum(), 1e-12)
    return iou.mean(), iou, acc
