"""Accuracy-level evidence for bf16 storage: a 30-step loss TRAJECTORY of the GPU step
(bf16 activations / MFMA operands, fp32 accumulation and master weights) against the fp32 CPU oracle step from identical
well-conditioned weights on a fixed small batch that both over-fit (the dense loss falls by ~60 % over the 30 steps).
Asserted per step: every loss within 2 % of the oracle's at the same step for the first 15 steps and within 5 % up to step 30
(measured: <= 2 % through step ~20, worst 3.2 % at step 27 of the pixel-distillation run; InfoNCE 5 % / 8 %, measured worst
5.3 % at step 22: two Adam trajectories from bf16- and fp32-rounded gradients drift apart slowly, they do not diverge);
at the end: every trained tensor's cosine with the oracle's >= 0.998 (measured minimum 0.9990, all but one >= 0.9997).  Also reported (and
bounded from below): the cosine of the accumulated UPDATE w_30 - w_0.  AdamW normalises every element's step to ~lr whatever
the gradient's size, so elements whose gradient is rounding noise on either side move by a random +-lr: the update cosine
(measured 0.67-0.71 on the large decoder tensors, 0.94-0.99 on the small ones, norm-weighted 0.75) is a much harsher number
than the loss trajectory or the weights themselves, and is what bf16 storage costs at lr = 1e-4 on near-zero gradients."""
import numpy as np
import pytest
import torch

from oracle.step import OracleStep
from tests.synth import damp_residual, fill_by_name

pytestmark = pytest.mark.gpu
STEPS = 30


def _cos(a, b):
    a, b = a.double().ravel(), b.double().ravel()
    return float(a @ b / (a.norm() * b.norm() + 1e-30))


@pytest.mark.parametrize("contr", [False, True])
def test_thirty_step_loss_trajectory_tracks_fp32_oracle(contr):
    from openess_amd.training.pretrain_step import PretrainStep
    torch.manual_seed(3)
    B, H, W, nwin = 2, 64, 96, 3
    st = PretrainStep(config_option="frame2voxel", img_size=(H, W), nr_events_data=nwin, if_spatial_contrastive=contr,
                      superpixel_size=25, lr=1e-4)
    ref = OracleStep("frame2voxel", 11, nwin, 5, contr, 25, lr=1e-4)
    for name, m in st.models_dict.items():
        fill_by_name(m, 100 + len(name))
        fill_by_name(ref.modules()[name], 100 + len(name), sorted(m.state_dict().keys()))
        damp_residual(m), damp_residual(ref.modules()[name])
    w0 = {f"{k}.{n}": p.detach().float().cpu().clone() for k, m in st.models_dict.items() for n, p in m.named_parameters()
          if p.requires_grad}
    g = torch.Generator().manual_seed(8)
    ev = (torch.randn(B, nwin * 5, H, W, generator=g) * (torch.rand(B, nwin * 5, H, W, generator=g) > 0.7)).contiguous()
    frame = torch.rand(B, 3, H, W, generator=g)
    pl = torch.randint(0, 11, (B, H // 8, W // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)   # learnable blocks
    sp = torch.randint(0, 25, (B, H // 8, W // 8), generator=g).repeat_interleave(8, 1).repeat_interleave(8, 2)
    S = int((sp + torch.arange(B)[:, None, None] * 25).max()) + 1
    dev_batch = (ev.cuda(), None, frame.cuda(), pl.cuda(), sp.cuda(), S)
    traj, worst = [], {}
    for it in range(STEPS):
        losses, _, _ = st.train_step(dev_batch)
        lref, _ = ref.train_step((ev, None, frame, pl, sp))
        row = {}
        for k in lref:
            a, b = float(losses[k]), float(lref[k])
            row[k] = (a, b)
            rel = abs(a - b) / abs(b)
            worst[k] = max(worst.get(k, 0.0), rel)
            early = it < 15
            bound = (5e-2 if early else 8e-2) if k == "contrastive_nce_loss" else (2e-2 if early else 5e-2)
            assert rel <= bound, (it, k, a, b)
        traj.append(row)
    first, last = traj[0]["dense_clip_loss"][1], traj[-1]["dense_clip_loss"][1]
    assert last < first                                   # the oracle actually trains on this batch (the comparison is not vacuous)
    print("trajectory worst relative loss error:", {k: round(v, 4) for k, v in worst.items()},
          "dense loss", round(first, 4), "->", round(last, 4))
    refp = {f"{k}.{n}": p.detach() for k, m in ref.modules().items() for n, p in m.named_parameters() if p.requires_grad}
    wc, uc, wts = [], [], []
    for k, m in st.models_dict.items():
        for n, p in m.named_parameters():
            key = f"{k}.{n}"
            if not p.requires_grad or key not in refp or p.grad is None:
                continue
            a, b = p.detach().float().cpu(), refp[key]
            wc.append(_cos(a, b))
            da, db = a - w0[key], b - w0[key]
            if float(db.norm()) > 1e-3 * float(b.norm()) and not key.endswith(("model.0.bias", "model.3.bias")):
                uc.append(_cos(da, db))                    # (conv bias in front of an affine-free InstanceNorm has a zero true gradient)
                wts.append(float(db.norm()))
    assert min(wc) >= 0.998, min(wc)          # measured: 0.9990 (one small tensor), every other tensor >= 0.9997
    uc, wts = np.array(uc), np.array(wts)
    print(f"weight cosine min {min(wc):.5f}; update cosine: norm-weighted mean {float((uc * wts).sum() / wts.sum()):.4f}, "
          f"median {float(np.median(uc)):.4f}, min {float(uc.min()):.4f} over {len(uc)} tensors")
    assert float((uc * wts).sum() / wts.sum()) >= 0.6
