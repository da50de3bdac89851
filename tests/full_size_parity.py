"""Full-size step parity (test infrastructure; imported by tests/ and by bench.py's cpu_baseline leg only): ONE sample at the
BASELINE geometry (440 x 640 crop, 5-bin sub-windows) through the HIP pre-training step (openess_amd.training.pretrain_step) and
through the CPU oracle (oracle/step.py restates training/pretrain_trainer.py:364-534) from identical, well-conditioned weights:
Dice + CE loss, per-pixel argmax of the student logits and (optionally) the superpixel InfoNCE.  No optimiser step on either side."""
import numpy as np
import torch


def compare(event_voxels, frame, pl, sp=None, nwin=20, bins=5, superpixel_size=100, seed=3):
    """event_voxels: fp32 [1, nwin*bins, H, W] (CPU); frame [1, 3, H, W]; pl int64 [1, H, W]; sp int64 [1, H, W] or None.
    Returns dict(loss_hip, loss_oracle, rel, argmax_agree[, nce_hip, nce_oracle, nce_rel])."""
    from openess_amd.training.pretrain_step import PretrainStep
    from oracle import losses as ol
    from oracle import nets as on
    from oracle.step import OracleStep
    from tests import synth
    contr = sp is not None
    H, W = frame.shape[-2:]
    torch.manual_seed(seed)
    st = PretrainStep(config_option="frame2voxel", img_size=(H, W), nr_events_data=nwin, nr_temporal_bins=bins,
                      if_spatial_contrastive=contr, superpixel_size=superpixel_size, lr=1e-4)
    ref = OracleStep("frame2voxel", 11, nwin, bins, contr, superpixel_size, lr=1e-4)
    for name, m in st.models_dict.items():
        synth.fill_by_name(m, 100 + len(name))
        synth.fill_by_name(ref.modules()[name], 100 + len(name), sorted(m.state_dict().keys()))
        synth.damp_residual(m), synth.damp_residual(ref.modules()[name])
    dev = "cuda"
    out = {}
    with torch.no_grad():
        # HIP side: frozen half (recurrent encoder, teacher) + decoder forward + loss
        batch = (event_voxels.to(dev), None, frame.to(dev), pl.to(dev)) + ((sp.to(dev), int(sp.max()) + 1) if contr else ())
        h = st.front(batch)
        st._set_modes()
        st._join_front(h)
        pred, feat_voxel = st.task_backend(h.content)
        logits_hip = pred[1].float()
        out["loss_hip"] = float(st.task_loss(pred[1], batch[3]))
        if contr:
            feat_frame = st.model_frame.head(h.teacher_enc)
            out["nce_hip"] = float(st.nce_loss(st._pool(feat_voxel, batch[4], batch[5]), st._pool(feat_frame, batch[4], batch[5])))
        torch.cuda.synchronize()
        # oracle side (oracle/step.py: loss(), without the backward)
        ref.model_frame.train()
        ref.back_end.train()
        states = None
        for i in range(nwin):
            x = on.event_preprocess(event_voxels[:, i * bins:(i + 1) * bins])
            _, states, latent = ref.front(x, states)
        pred_o, feat_voxel_o = ref.back_end({k: v for k, v in latent.items()})
        out["loss_oracle"] = float(ol.task_loss(pred_o[1], pl, 11))
        if contr:
            feat_frame_o = ref.model_frame(frame)
            out["nce_oracle"] = float(ol.nce_loss(ol.superpixel_pool(feat_voxel_o, sp, superpixel_size), ol.superpixel_pool(feat_frame_o, sp, superpixel_size)))
            out["nce_rel"] = abs(out["nce_hip"] - out["nce_oracle"]) / abs(out["nce_oracle"])
    out["rel"] = abs(out["loss_hip"] - out["loss_oracle"]) / abs(out["loss_oracle"])
    a, b = logits_hip.argmax(1).cpu(), pred_o[1].argmax(1)
    out["argmax_agree"] = float((a == b).float().mean())
    # The logits themselves: rms error relative to their spread.  The argmax can only be expected to agree where the oracle's
    # top-2 margin is above that error (random-init weights: many near-ties), so agreement is reported on the pixels whose margin
    # exceeds 4 x the rms logit error, together with the share of such pixels.
    lo = pred_o[1]
    err = (logits_hip.cpu() - lo)
    err_rms = float(err.pow(2).mean().sqrt())
    out["logit_rel_rms_err"] = err_rms / float(lo.std())
    top2 = lo.topk(2, dim=1).values
    margin = top2[:, 0] - top2[:, 1]
    clear = margin > 4.0 * err_rms
    out["argmax_agree_clear_margin"] = float((a == b)[clear].float().mean()) if bool(clear.any()) else 1.0
    out["clear_margin_pixels"] = float(clear.float().mean())
    out["median_margin_over_rms_err"] = float(margin.median()) / err_rms
    return {k: (round(v, 6) if isinstance(v, float) else v) for k, v in out.items()}


def synthetic_sample(nwin, n_per, H_sensor=480, W=640, crop=40, bins=5, seed=1205, contrastive=True):
    """One synthetic DSEC-shaped sample voxelised by the oracle's C port / NumPy voxelizer (CPU)."""
    from openess_amd.datasets import _synth
    from oracle import events as oe
    x, y, t, p = _synth.dsec_raw_events(nwin * n_per, H_sensor, W, seed=seed)
    rmap = _synth.rectify_map(H_sensor, W)
    try:
        from oracle import cport
        ev = cport.dsec_event_tensor(x, y, t, p, rmap, nwin, bins, H_sensor, W, crop)
    except Exception:
        ev = oe.dsec_event_tensor(x, y, t, p, rmap, nwin, bins, H_sensor, W, crop)
    ev = torch.from_numpy(np.ascontiguousarray(ev))[None]
    Hn = H_sensor - crop
    g = torch.Generator().manual_seed(5)
    frame = torch.rand(1, 3, Hn, W, generator=g)
    pl = torch.randint(0, 11, (1, Hn, W), generator=g)
    yy = (torch.arange(Hn) * 10 // Hn)[:, None]
    xx = (torch.arange(W) * 10 // W)[None, :]
    sp = (yy * 10 + xx)[None].long() if contrastive else None
    return ev, frame, pl, sp
