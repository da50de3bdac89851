"""Pin the oracle against golden vectors produced by the reference itself
(tests/golden/gen_golden.py, run in the build container).  CPU only."""
import os

import numpy as np
import pytest
import torch

from oracle import events as oe
from oracle import losses as ol


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_trilinear_matches_reference(golden_events, tag):
    g = golden_events
    C, H, W = g[f"tri_{tag}_chw"]
    out = oe.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W)
    ref = g[f"tri_{tag}_out_norm0"]
    # identical sequential f32 accumulation order on CPU -> bit-exact
    assert np.array_equal(out, ref)
    outn = oe.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W,
                                  normalize=True)
    np.testing.assert_allclose(outn, g[f"tri_{tag}_out_norm1"], rtol=1e-4, atol=1e-5)


def test_trilinear_integer_coords_exact(golden_events):
    g = golden_events
    C, H, W = g["tri_int_chw"]
    out = oe.voxelgrid_trilinear(g["tri_int_x"], g["tri_int_y"], g["tri_int_p"], g["tri_int_t"], C, H, W)
    assert np.array_equal(out, g["tri_int_out"])
    assert np.array_equal(out, np.round(out))          # pure +-1 counts


@pytest.mark.parametrize("bins", [5, 2, 1])
@pytest.mark.parametrize("sp", [0, 1])
def test_nearest_matches_reference(golden_events, bins, sp):
    g = golden_events
    H, W = g["near_hw"]
    out = oe.voxelgrid_nearest(g["near_ev"], (H, W), bins, bool(sp))
    assert np.array_equal(out, g[f"near_out_b{bins}_sp{sp}"])


def test_nearest_edge_cases(golden_events):
    g = golden_events
    H, W = g["near_hw"]
    assert np.array_equal(oe.voxelgrid_nearest(g["near_evf"], (H, W), 5, False), g["near_outf_b5_sp0"])
    assert np.array_equal(oe.voxelgrid_nearest(g["near_ev0"], (H, W), 5, False), g["near_out0_b5_sp0"])
    assert np.array_equal(oe.voxelgrid_nearest(g["near_ev1"], (H, W), 5, True), g["near_out1_b5_sp1"])
    with pytest.raises(IndexError):
        oe.voxelgrid_nearest(np.zeros((0, 4), np.int64), (H, W), 5, True)


def test_histogram_and_normalize(golden_events):
    g = golden_events
    H, W = g["near_hw"]
    assert np.array_equal(oe.event_histogram(g["hist_ev"], (H, W)), g["hist_out"])
    assert np.array_equal(oe.generate_input_representation(g["hist_ev"], "histogram", (H, W)), g["hist_out"])
    np.testing.assert_allclose(oe.masked_normalize(g["norm_in"]), g["norm_out"], rtol=2e-5, atol=2e-6)
    assert np.array_equal(oe.masked_normalize(np.zeros((2, 3, 4), np.float32)), g["norm_zero_out"])


def test_e2vid_voxel_grid(golden_events):
    g = golden_events
    H, W = g["near_hw"]
    assert np.array_equal(oe.e2vid_voxel_grid(g["e2v_ev"], 5, W, H), g["e2v_out"])


@pytest.mark.parametrize("tag,K", [("a", 11), ("b", 6)])
def test_task_and_dice_loss(golden_losses, tag, K):
    g = golden_losses
    lg = torch.from_numpy(g[f"task_{tag}_logits"]).requires_grad_(True)
    tgt = torch.from_numpy(g[f"task_{tag}_target"])
    loss = ol.task_loss(lg, tgt, K)
    loss.backward()
    np.testing.assert_allclose(loss.item(), g[f"task_{tag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(lg.grad.numpy(), g[f"task_{tag}_grad"], rtol=1e-5, atol=1e-9)
    lg2 = torch.from_numpy(g[f"task_{tag}_logits"]).requires_grad_(True)
    d = ol.dice_loss(lg2, tgt, K)
    d.backward()
    np.testing.assert_allclose(d.item(), g[f"dice_{tag}_loss"], rtol=1e-6)
    np.testing.assert_allclose(lg2.grad.numpy(), g[f"dice_{tag}_grad"], rtol=1e-5, atol=1e-9)


def test_nce(golden_losses):
    g = golden_losses
    k = torch.from_numpy(g["nce_k"]).requires_grad_(True)
    q = torch.from_numpy(g["nce_q"]).requires_grad_(True)
    l = ol.nce_loss(k, q, 0.07)
    l.backward()
    np.testing.assert_allclose(l.item(), g["nce_loss"], rtol=1e-6)
    np.testing.assert_allclose(k.grad.numpy(), g["nce_gk"], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(q.grad.numpy(), g["nce_gq"], rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_superpixel_pool(golden_losses, tag):
    g = golden_losses
    fk = torch.from_numpy(g[f"sp_{tag}_feat_k"]).requires_grad_(True)
    fq = torch.from_numpy(g[f"sp_{tag}_feat_q"]).requires_grad_(True)
    ids = torch.from_numpy(g[f"sp_{tag}_ids"])
    k = ol.superpixel_pool(fk, ids, int(g[f"sp_{tag}_size"]))
    q = ol.superpixel_pool(fq, ids, int(g[f"sp_{tag}_size"]))
    assert k.shape == g[f"sp_{tag}_k"].shape
    np.testing.assert_allclose(k.detach().numpy(), g[f"sp_{tag}_k"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(q.detach().numpy(), g[f"sp_{tag}_q"], rtol=1e-5, atol=1e-6)
    w = torch.from_numpy(g[f"sp_{tag}_w"])
    ((k * w).sum() + (q * w.flip(0)).sum()).backward()
    np.testing.assert_allclose(fk.grad.numpy(), g[f"sp_{tag}_gk"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(fq.grad.numpy(), g[f"sp_{tag}_gq"], rtol=1e-5, atol=1e-7)


def test_metrics(golden_losses):
    g = golden_losses
    cm = sum(ol.confusion_matrix(p, t, 11) for p, t in zip(g["met_pred"], g["met_gt"]))
    assert np.array_equal(cm, g["met_cm"])
    miou, _, acc = ol.miou_acc(cm)
    assert miou == pytest.approx(float(g["met_miou"]), rel=1e-12)
    assert acc == pytest.approx(float(g["met_acc"]), rel=1e-12)


def test_c_port_matches_reference_golden(golden_events):
    """oracle/voxel_oracle.c (scalar C port, used as bench.py's CPU baseline) against the same golden vectors."""
    from oracle import cport
    g = golden_events
    for tag in ("a", "b", "c"):
        C, H, W = (int(v) for v in g[f"tri_{tag}_chw"])
        out = cport.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W)
        np.testing.assert_allclose(out, g[f"tri_{tag}_out_norm0"], rtol=0, atol=2e-6)      # event-major vs 8-pass order
        cnt = cport.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W, True)
        ref = oe.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W, count_mode=True)
        assert np.array_equal(cnt, ref)
    C, H, W = (int(v) for v in g["tri_int_chw"])
    assert np.array_equal(cport.voxelgrid_trilinear(g["tri_int_x"], g["tri_int_y"], g["tri_int_p"], g["tri_int_t"], C, H, W), g["tri_int_out"])
    H, W = (int(v) for v in g["near_hw"])
    for bins in (5, 2, 1):
        for sp in (0, 1):
            assert np.array_equal(cport.voxelgrid_nearest_i64(g["near_ev"], (H, W), bins, bool(sp)), g[f"near_out_b{bins}_sp{sp}"])
    assert np.array_equal(cport.voxelgrid_nearest_i64(g["near_ev0"], (H, W), 5, False), g["near_out0_b5_sp0"])


def test_c_port_dsec_sample_matches_numpy_oracle():
    from oracle import cport
    from tests import synth
    C, H, W, crop, nwin, n_per = 5, 96, 128, 8, 3, 4000
    x, y, t, p = synth.dsec_raw_events(nwin * n_per, H, W, seed=3)
    rm = synth.rectify_map(H, W)
    a = cport.dsec_event_tensor(x, y, t, p, rm, nwin, C, H, W, crop)
    b = oe.dsec_event_tensor(x, y, t, p, rm, nwin, C, H, W, crop)
    np.testing.assert_allclose(a, b, rtol=0, atol=2e-6)
    assert np.array_equal(cport.dsec_event_tensor(x, y, t, p, rm, nwin, C, H, W, crop, True),
                          oe.dsec_event_tensor(x, y, t, p, rm, nwin, C, H, W, crop, count_mode=True))


def test_consistency_losses_golden():
    """a16: oracle restatement vs the torch calls the reference makes (tests/golden/gen_golden_consistency.py)."""
    import os
    import torch
    from oracle import losses as ol
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "consistency.npz"))
    for tag in ("feat", "logit"):
        a, b = torch.from_numpy(g[f"{tag}_a"]), torch.from_numpy(g[f"{tag}_b"])
        l1, lc = ol.consistency_losses(a, b, a, b)
        np.testing.assert_allclose(l1.numpy(), g[f"{tag}_l1"], rtol=1e-6)
        np.testing.assert_allclose(lc.numpy(), g[f"{tag}_cos"], rtol=1e-6, atol=1e-7)


# ---- the reference's own EventPreprocessor / CropParameters (tests/golden/gen_golden_e2vid_pre.py)
@pytest.fixture(scope="module")
def golden_pre():
    return dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "e2vid_pre.npz")))


@pytest.mark.parametrize("case", ["sparse", "dense", "zeros", "single"])
def test_event_preprocessor_oracle_equals_reference(golden_pre, case):
    """oracle.nets.event_preprocess against EventPreprocessor.__call__ (e2vid/utils/inference_utils.py:70-87) run from the
    reference: bit-exact, including the unguarded 0/0 = NaN of a tensor with a single non-zero value and the untouched
    all-zero tensor (`if num_nonzeros > 0`)."""
    import torch
    from oracle import nets as on
    got = on.event_preprocess(torch.from_numpy(golden_pre[f"pre_in_{case}"].copy())).numpy()
    np.testing.assert_array_equal(got, golden_pre[f"pre_out_{case}"])


def test_crop_parameters_mirror_equals_reference(golden_pre):
    """CropParameters (e2vid/utils/inference_utils.py:284-311): every derived attribute and the reflection pad."""
    import torch
    from openess_amd.e2vid.utils.inference_utils import CropParameters
    attrs = ("width_crop_size", "height_crop_size", "padding_top", "padding_bottom", "padding_left", "padding_right", "cx", "cy",
             "ix0", "ix1", "iy0", "iy1")
    for (w, h, n), want in zip(golden_pre["crop_cases"].tolist(), golden_pre["crop_attrs"].tolist()):
        cp = CropParameters(w, h, n)
        assert [getattr(cp, a) for a in attrs] == want, (w, h, n)
        assert cp.needs_pad == any(want[2:6])
    cp = CropParameters(44, 30, 3)
    np.testing.assert_array_equal(cp.pad(torch.from_numpy(golden_pre["pad_in"])).numpy(), golden_pre["pad_out"])
