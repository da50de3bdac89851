"""Golden vectors for the a16 consistency losses (training/openess_trainer.py:497-503).  The reference computes them
with two library calls, `torch.nn.L1Loss()(feat_a, feat_b)` and `torch.mean(1 - F.cosine_similarity(la, lb, dim=1))`;
the arithmetic lives in PyTorch (third party, pinned torch 2.1 in the reference's INSTALL.md), so the vectors are those
exact calls run in this container (torch 2.10 CPU, fp32) with their input gradients.
    python tests/golden/gen_golden_consistency.py   ->  tests/golden/consistency.npz"""
import os
import numpy as np
import torch
import torch.nn.functional as F

torch.manual_seed(1205)
out = {}
for tag, (B, C, H, W) in {"feat": (2, 64, 7, 9), "logit": (2, 11, 13, 10)}.items():
    a = torch.randn(B, C, H, W, requires_grad=True)
    b = torch.randn(B, C, H, W, requires_grad=True)
    if tag == "logit":
        with torch.no_grad():
            b[0, :, 0, 0] = 0.0                     # zero vector: the per-norm eps clamp path
            a[1, :, 2, 3] = b[1, :, 2, 3]           # identical vectors: cos = 1
            a[0, 0, 1, 1] = b[0, 0, 1, 1]
    l1 = torch.nn.L1Loss()(a, b)
    ga, gb = torch.autograd.grad(l1, (a, b))
    out[f"{tag}_a"], out[f"{tag}_b"] = a.detach().numpy(), b.detach().numpy()
    out[f"{tag}_l1"], out[f"{tag}_l1_ga"], out[f"{tag}_l1_gb"] = l1.detach().numpy(), ga.numpy(), gb.numpy()
    lc = torch.mean(1 - F.cosine_similarity(a, b, dim=1))
    ga, gb = torch.autograd.grad(lc, (a, b))
    out[f"{tag}_cos"], out[f"{tag}_cos_ga"], out[f"{tag}_cos_gb"] = lc.detach().numpy(), ga.numpy(), gb.numpy()
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "consistency.npz"), **out)
print({k: v.shape for k, v in out.items()})
