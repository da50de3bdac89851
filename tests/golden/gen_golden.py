#!/usr/bin/env python3
"""Generate golden vectors by RUNNING THE REFERENCE (imported from /root/reference).

Runs only in the build container (the reference never travels to the GPU box).  Writes
small .npz fixtures (inputs + expected outputs, no code) next to this script:

    python tests/golden/gen_golden.py            # all groups
    python tests/golden/gen_golden.py events     # one group

Import tricks follow SURVEY.md 8c: `datasets/` is shadowed by the HuggingFace package so
data_util.py is loaded by path; `models/__init__.py` star-imports torchvision/mmcv users so a
bare package is pre-registered; torchvision is stubbed (only the unused StyleEncoderE2VID
touches it).
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
torch.set_num_threads(1)


def _load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ref_models():
    if "models" not in sys.modules or not hasattr(sys.modules["models"], "__path__"):
        pkg = types.ModuleType("models")
        pkg.__path__ = [os.path.join(REF, "models")]
        sys.modules["models"] = pkg
    for stub in ("torchvision", "torchvision.models"):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    import models.style_networks as sn
    import models.deeplabv3 as dl
    import models._resnet as rn
    return sn, dl, rn


def seeded_state(module, seed):
    """Deterministic weights keyed by parameter NAME (so the product model, which keeps the
    reference's state_dict keys, can be filled identically without shipping weight files)."""
    sd = module.state_dict()
    out = {}
    for i, k in enumerate(sorted(sd.keys())):
        v = sd[k]
        rng = np.random.default_rng([seed, i])
        if not v.dtype.is_floating_point:
            out[k] = v.clone()
            continue
        shp = tuple(v.shape)
        if k.endswith("running_var"):
            a = rng.uniform(0.5, 1.5, shp)
        elif k.endswith("running_mean"):
            a = rng.normal(0, 0.1, shp)
        elif v.ndim >= 2:
            fan_in = int(np.prod(shp[1:]))
            a = rng.normal(0, 1.0 / np.sqrt(fan_in), shp)
        elif k.endswith("weight"):
            a = rng.uniform(0.5, 1.5, shp)
        else:
            a = rng.normal(0, 0.1, shp)
        out[k] = torch.from_numpy(np.asarray(a, dtype=np.float32)).reshape(v.shape)
    return out


# ------------------------------------------------------------------------------------ events
def gen_events():
    from DSEC.dataset.representations import VoxelGrid
    du = _load_by_path("ref_data_util", os.path.join(REF, "datasets/data_util.py"))
    rng = np.random.default_rng(1205)
    out = {}

    # G2: VoxelGrid.convert, fractional coords incl. out-of-range (-2, W+1)
    for tag, (C, H, W, N) in {"a": (5, 24, 32, 3000), "b": (2, 17, 23, 1500), "c": (3, 8, 8, 64)}.items():
        x = rng.uniform(-2, W + 1, N).astype(np.float32)
        y = rng.uniform(-2, H + 1, N).astype(np.float32)
        p = rng.integers(0, 2, N).astype(np.float32)
        t = np.sort(rng.integers(0, 50000, N)).astype(np.float64)
        t[0], t[-1] = 0, 50000
        tf = (t - t[0]).astype(np.float32)
        tf = tf / tf[-1]
        for norm in (False, True):
            vg = VoxelGrid(C, H, W, normalize=norm)
            g = vg.convert(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(tf))
            out[f"tri_{tag}_out_norm{int(norm)}"] = g.numpy()
        out[f"tri_{tag}_x"], out[f"tri_{tag}_y"], out[f"tri_{tag}_p"], out[f"tri_{tag}_t"] = x, y, p, tf
        out[f"tri_{tag}_chw"] = np.array([C, H, W])
    # integer-coordinate case: every weight is exactly +-1 or 0 -> exact integer histogram
    C, H, W, N = 5, 16, 20, 4000
    x = rng.integers(-1, W + 1, N).astype(np.float32)
    y = rng.integers(-1, H + 1, N).astype(np.float32)
    p = rng.integers(0, 2, N).astype(np.float32)
    tq = np.sort(rng.integers(0, C, N)).astype(np.float32)
    tq[0], tq[-1] = 0, C - 1
    tf = tq / np.float32(C - 1)          # t_norm = (C-1)*tf is an exact integer for C-1 = 4
    vg = VoxelGrid(C, H, W, normalize=False)
    g = vg.convert(torch.from_numpy(x), torch.from_numpy(y), torch.from_numpy(p), torch.from_numpy(tf))
    out["tri_int_x"], out["tri_int_y"], out["tri_int_p"], out["tri_int_t"] = x, y, p, tf
    out["tri_int_chw"] = np.array([C, H, W])
    out["tri_int_out"] = g.numpy()

    # G1: generate_voxel_grid (DDD17 flavour), int64 events (x, y, t, p)
    H, W, N = 26, 35, 3000
    ev = np.stack([rng.integers(-2, W + 2, N), rng.integers(-2, H + 2, N),
                   np.sort(rng.integers(10**6, 10**6 + 50000, N)), rng.integers(0, 2, N)], -1).astype(np.int64)
    out["near_ev"] = ev
    out["near_hw"] = np.array([H, W])
    for bins in (5, 2, 1):
        for sp in (False, True):
            out[f"near_out_b{bins}_sp{int(sp)}"] = du.generate_voxel_grid(ev.copy(), (H, W), bins, sp)
    # polarity given as +-1 and float64 events
    evf = ev.astype(np.float64)
    evf[:, 3] = 2 * evf[:, 3] - 1
    evf[:, 2] += rng.uniform(0, 1, N)
    evf[:, 2] = np.sort(evf[:, 2])
    out["near_evf"] = evf
    out["near_outf_b5_sp0"] = du.generate_voxel_grid(evf.copy(), (H, W), 5, False)
    # deltaT == 0 (all timestamps equal)
    ev0 = ev[:200].copy()
    ev0[:, 2] = 777
    out["near_ev0"] = ev0
    out["near_out0_b5_sp0"] = du.generate_voxel_grid(ev0.copy(), (H, W), 5, False)
    # single event
    out["near_ev1"] = ev[5:6].copy()
    out["near_out1_b5_sp1"] = du.generate_voxel_grid(ev[5:6].copy(), (H, W), 5, True)

    # G1c: histogram (in-range coordinates only; the reference has no bounds check)
    evh = ev.copy()
    evh[:, 0] = np.clip(evh[:, 0], 0, W - 1)
    evh[:, 1] = np.clip(evh[:, 1], 0, H - 1)
    out["hist_ev"] = evh
    out["hist_out"] = du.generate_event_histogram(evh.copy(), (H, W))

    # G3: normalize_voxel_grid
    v = torch.from_numpy(out["near_out_b5_sp0"].copy())
    out["norm_in"] = v.numpy().copy()
    out["norm_out"] = du.normalize_voxel_grid(v).numpy()
    out["norm_zero_out"] = du.normalize_voxel_grid(torch.zeros(2, 3, 4)).numpy()

    # a6: e2vid events_to_voxel_grid (columns t,x,y,p; in-range coords).  np.int was removed from
    # NumPy >= 1.24; the reference line `astype(np.int)` needs the alias to run unmodified.
    if not hasattr(np, "int"):
        np.int = int
    for stub in ("cv2", "albumentations"):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)
    try:
        import e2vid.utils.inference_utils as iu
        eve = np.stack([evf[:, 2], np.clip(evf[:, 0], 0, W - 1), np.clip(evf[:, 1], 0, H - 1),
                        (evf[:, 3] > 0).astype(np.float64)], -1)
        out["e2v_ev"] = eve
        out["e2v_out"] = iu.events_to_voxel_grid(eve.copy(), 5, W, H)
    except Exception as e:  # pragma: no cover
        print("e2vid inference_utils not importable:", repr(e))
    np.savez_compressed(os.path.join(HERE, "events.npz"), **out)
    print("events.npz:", len(out), "arrays")


# ------------------------------------------------------------------------------------ losses
def gen_losses():
    from utils.loss_functions import TaskLoss, DiceLoss, NCELoss
    from evaluation.metrics import MetricsSemseg
    torch.manual_seed(1205)
    rng = np.random.default_rng(1206)
    out = {}
    for tag, (B, K, H, W) in {"a": (2, 11, 12, 16), "b": (3, 6, 9, 7)}.items():
        logits = torch.randn(B, K, H, W) * 2
        tgt = torch.from_numpy(rng.integers(0, K, (B, H, W))).long()
        ign = torch.from_numpy(rng.uniform(0, 1, (B, H, W)) < 0.1)
        tgt[ign] = 255
        lg = logits.clone().requires_grad_(True)
        loss = TaskLoss(losses=["dice", "cross_entropy"], num_classes=K, ignore_index=255)(lg, tgt)
        loss.backward()
        out[f"task_{tag}_logits"], out[f"task_{tag}_target"] = logits.numpy(), tgt.numpy()
        out[f"task_{tag}_loss"], out[f"task_{tag}_grad"] = loss.detach().numpy(), lg.grad.numpy()
        lg2 = logits.clone().requires_grad_(True)
        d = DiceLoss(num_classes=K, ignore_index=255)(lg2, tgt)
        d.backward()
        out[f"dice_{tag}_loss"], out[f"dice_{tag}_grad"] = d.detach().numpy(), lg2.grad.numpy()
    S, Cc = 37, 16
    k = torch.randn(S, Cc, requires_grad=True)
    q = torch.randn(S, Cc, requires_grad=True)
    l = NCELoss(temperature=0.07)(k, q)
    l.backward()
    out["nce_k"], out["nce_q"], out["nce_loss"] = k.detach().numpy(), q.detach().numpy(), l.detach().numpy()
    out["nce_gk"], out["nce_gq"] = k.grad.numpy(), q.grad.numpy()

    # G5 superpixel pooling: execute the reference's inline lines (pretrain_trainer.py:445-465)
    src = open(os.path.join(REF, "training/pretrain_trainer.py")).read().split("\n")
    block = src[445:465]
    ind = len(block[0]) - len(block[0].lstrip())
    code = "\n".join(line[ind:] for line in block)
    for tag, (B, Cc, H, W, sps, maxid) in {"a": (2, 8, 10, 12, 25, 25), "b": (3, 4, 6, 5, 10, 14)}.items():
        feat_voxel = torch.randn(B, Cc, H, W, requires_grad=True)
        feat_frame = torch.randn(B, Cc, H, W, requires_grad=True)
        sp = torch.from_numpy(rng.integers(0, maxid, (B, H, W))).long()
        if tag == "a":
            sp[sp == 3] = 4            # an empty segment
        settings = types.SimpleNamespace(superpixel_size=sps)
        env = {"torch": torch, "feat_voxel": feat_voxel, "feat_frame": feat_frame, "superpixels": sp.clone(),
               "self": types.SimpleNamespace(settings=settings)}
        exec(code, env)
        kk, qq = env["k"], env["q"]
        w = torch.randn_like(kk)
        ((kk * w).sum() + (qq * w.flip(0)).sum()).backward()
        out[f"sp_{tag}_feat_k"], out[f"sp_{tag}_feat_q"] = feat_voxel.detach().numpy(), feat_frame.detach().numpy()
        out[f"sp_{tag}_ids"], out[f"sp_{tag}_size"] = sp.numpy(), np.array(sps)
        out[f"sp_{tag}_k"], out[f"sp_{tag}_q"], out[f"sp_{tag}_w"] = kk.detach().numpy(), qq.detach().numpy(), w.numpy()
        out[f"sp_{tag}_gk"], out[f"sp_{tag}_gq"] = feat_voxel.grad.numpy(), feat_frame.grad.numpy()

    # G6 metrics
    K = 11
    m = MetricsSemseg(K, 255, [str(i) for i in range(K)])
    preds, gts = [], []
    for _ in range(3):
        pred = torch.from_numpy(rng.integers(0, K - 2, (2, 9, 13))).long()     # classes 9,10 never predicted
        gt = torch.from_numpy(rng.integers(0, K - 1, (2, 9, 13))).long()       # class 10 absent
        gt[torch.from_numpy(rng.uniform(0, 1, (2, 9, 13)) < 0.15)] = 255
        m.update_batch(pred, gt)
        preds.append(pred.numpy())
        gts.append(gt.numpy())
    s = m.get_metrics_summary()
    out["met_pred"], out["met_gt"] = np.stack(preds), np.stack(gts)
    out["met_cm"], out["met_miou"], out["met_acc"] = s["cm"].numpy(), np.array(float(s["miou"])), np.array(float(s["acc"]))
    np.savez_compressed(os.path.join(HERE, "losses.npz"), **out)
    print("losses.npz:", len(out), "arrays")


GROUPS = {"events": gen_events, "losses": gen_losses}

if __name__ == "__main__":
    try:
        import gen_golden_nets  # noqa: F401  (registers more groups)
        GROUPS.update(gen_golden_nets.GROUPS)
    except ImportError:
        pass
    def _datasets():       # own process: its import stubs (h5py, cv2, torchvision, `datasets`) must not leak into the other groups
        import subprocess
        subprocess.run([sys.executable, os.path.join(HERE, "gen_golden_datasets.py")], check=True)
    GROUPS["datasets"] = _datasets
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()
