#!/usr/bin/env python3
"""Golden vectors for the DATASET path (SURVEY 8a rows a2/a5, 8b dataset selector, 8f-2): runs the REFERENCE's own
`DSECEvents(...)` (datasets/DSEC_events_loader.py -> DSEC/dataset/provider.py -> sequence_ov.py::Sequence.__getitem__,
DSEC/utils/eventslicer.py) and `DDD17Events(...)` (datasets/ddd17_events_loader.py) over the deterministic fake trees of
tests/synth_datasets.py, and stores ONLY checksums / compact samples of what they return.

    python tests/golden/gen_golden.py datasets

Absent third-party modules are replaced by import stubs so that the reference files import unmodified:
  h5py       -> a File look-alike over the unpacked .npy layout (tests/synth_datasets.py:write_unpacked_h5)
  hdf5plugin -> empty;  numba.jit -> identity decorator (the jitted scan then runs as plain Python)
  torchvision.transforms(.functional) -> adjust_brightness / adjust_contrast RAISE: golden augmentation seeds are chosen
                so that those two branches are not taken (flip + noise branches are)
  cv2        -> imread via PIL; resize(INTER_NEAREST) restated from OpenCV's documented rule src = floor(dst * src/dst).
                cv2 is only on the DDD17 label/pl/superpixel resize path: those three outputs are "parity unpinned"
                (stated in DESIGN.md); everything else below is the reference's own arithmetic.
"""
import os
import random
import sys
import tempfile
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from tests import synth_datasets as sd  # noqa: E402
from tests.synth import compact  # noqa: E402


class _FakeH5(dict):
    def __init__(self, path, mode='r'):
        d = str(path)[:-3] + "_h5"
        super().__init__()
        for f in sorted(os.listdir(d)):
            self[f[:-4]] = np.load(os.path.join(d, f), mmap_mode='r')

    def __getitem__(self, k):
        return dict.__getitem__(self, k.replace("/", "_"))

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False

    def close(self):
        pass


def _cv2_resize_nearest(img, dsize, interpolation=None):
    w, h = dsize
    H, W = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def install_stubs():
    from PIL import Image
    h5 = types.ModuleType("h5py")
    h5.File = _FakeH5
    sys.modules["h5py"] = h5
    sys.modules["hdf5plugin"] = types.ModuleType("hdf5plugin")
    nb = types.ModuleType("numba")
    nb.jit = lambda *a, **k: (lambda f: f)
    sys.modules["numba"] = nb
    cv2 = types.ModuleType("cv2")
    cv2.INTER_NEAREST, cv2.IMREAD_ANYDEPTH = 0, 2
    cv2.imread = lambda p, flag=0: np.array(Image.open(p))
    cv2.resize = _cv2_resize_nearest
    sys.modules["cv2"] = cv2

    def _refuse(*a, **k):
        raise RuntimeError("torchvision is absent: golden seeds must not take the brightness/contrast branches")
    tv, tt, tf = types.ModuleType("torchvision"), types.ModuleType("torchvision.transforms"), types.ModuleType("torchvision.transforms.functional")
    tf.adjust_brightness = tf.adjust_contrast = _refuse
    tt.functional, tt.Compose, tt.Grayscale, tt.ToTensor = tf, _refuse, _refuse, _refuse
    tv.transforms = tt
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tt, "torchvision.transforms.functional": tf})
    ds = types.ModuleType("datasets")           # the reference's datasets/ has no __init__.py and is shadowed by HuggingFace's
    ds.__path__ = [os.path.join(REF, "datasets")]
    sys.modules["datasets"] = ds
    ex = types.ModuleType("datasets.extract_data_tools")
    ex.__path__ = [os.path.join(REF, "datasets", "extract_data_tools")]
    sys.modules["datasets.extract_data_tools"] = ex
    mpl = types.ModuleType("matplotlib")
    sys.modules.setdefault("matplotlib", mpl)
    sys.modules.setdefault("matplotlib.pyplot", types.ModuleType("matplotlib.pyplot"))
    if REF not in sys.path:
        sys.path.insert(0, REF)
    sys.dont_write_bytecode = True


def crc(a):
    a = np.ascontiguousarray(a.numpy() if torch.is_tensor(a) else a)
    return np.array([zlib.crc32(a.tobytes()), a.size], dtype=np.int64)


def aug_seed(want_flip=True, want_noise=True):
    """First python-random seed whose 4 draws give: flip as wanted, NO brightness, NO contrast, noise as wanted."""
    for k in range(10000):
        random.seed(k)
        r = [random.random() for _ in range(4)]
        if (r[0] >= 0.5) == want_flip and r[1] < 0.5 and r[2] < 0.5 and (r[3] >= 0.5) == want_noise:
            return k
    raise RuntimeError


def record(out, tag, item, root, n_fields):
    """item = reference __getitem__ tuple.  Float tensors -> crc + compact; integer tensors -> crc; path -> relative."""
    for j in range(n_fields - 1):
        v = item[j]
        out[f"{tag}_f{j}_crc"] = crc(v)
        out[f"{tag}_f{j}_shape"] = np.array(v.shape)
        out[f"{tag}_f{j}_dtype"] = np.array(str(v.dtype))
        if v.dtype.is_floating_point and v.numel() > 1 and not (v.numel() == 256 * 64 * 64):
            sub, ssum, sabs = compact(v.numpy(), n=2048)
            out[f"{tag}_f{j}_sub"], out[f"{tag}_f{j}_sum"], out[f"{tag}_f{j}_abs"] = sub, ssum, sabs
    out[f"{tag}_path"] = np.array(os.path.relpath(item[n_fields - 1], root))


def gen_datasets():
    install_stubs()
    torch.set_num_threads(4)
    out = {}
    tmp = tempfile.mkdtemp(prefix="oess_golden_ds_")
    # ------------------------------------------------------------------ DSEC
    droot = sd.make_dsec_tree(os.path.join(tmp, "dsec"))
    from datasets.DSEC_events_loader import DSECEvents
    common = dict(nr_events_data=4, delta_t_per_data=20, nr_events_window=3000, event_representation='voxel_grid',
                  nr_bins_per_data=5, require_paired_data=False, separate_pol=False, normalize_event=False,
                  semseg_num_classes=11, pl_sources='pl_fcclip_rgb', if_sam_distillation=False)
    cases = {
        "dsec_train_f2v": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
        "dsec_train_f2v_slic": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_slic_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
        "dsec_train_f2r": dict(mode='train', config_option='frame2recon', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
        "dsec_train_r2v_fixdur": dict(mode='train', config_option='recon2voxel', superpixel_sources='', augmentation=False, fixed_duration=True, skip_ratio=1),
        "dsec_val_f2v": dict(mode='val', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
        "dsec_train_f2v_skip": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=2),
    }
    for tag, kw in cases.items():
        ds = DSECEvents(dsec_dir=droot, **common, **kw)
        out[f"{tag}_len"] = np.array(len(ds))
        assert ds.require_paired_data is False
        for i in range(len(ds)):
            record(out, f"{tag}_{i}", ds[i], droot, 7)
    # augmentation: flip + noise taken, brightness / contrast not (see module docstring)
    for tag, opt in (("dsec_aug_f2v", "frame2voxel"), ("dsec_aug_f2r", "frame2recon")):
        ds = DSECEvents(dsec_dir=droot, **common, mode='train', config_option=opt, superpixel_sources='sp_sam_rgb',
                        augmentation=True, fixed_duration=False, skip_ratio=1)
        k = aug_seed(True, True)
        out[f"{tag}_pyseed"] = np.array(k)
        random.seed(k)
        torch.manual_seed(99)
        record(out, f"{tag}_0", ds[1], droot, 7)
    # more events requested than exist before the frame: start_index = 0 branch (sequence_ov.py:287-290), remainder drop (:302)
    ds = DSECEvents(dsec_dir=droot, **dict(common, nr_events_window=20001, nr_events_data=3), mode='train', config_option='frame2voxel',
                    superpixel_sources='', augmentation=False, fixed_duration=False, skip_ratio=1)
    record(out, "dsec_short_0", ds[0], droot, 7)
    record(out, "dsec_short_1", ds[len(ds) - 1], droot, 7)

    # ------------------------------------------------------------------ DDD17
    root17 = sd.make_ddd17_tree(os.path.join(tmp, "ddd17"))
    from datasets.ddd17_events_loader import DDD17Events
    c17 = dict(event_representation='voxel_grid', nr_events_data=4, delta_t_per_data=50, nr_bins_per_data=5, require_paired_data=False,
               normalize_event=False, fixed_duration=False, nr_events_per_data=700, resize=True, random_crop=False,
               pl_sources='pl_fcclip_rgb', superpixel_sources='sp_sam_rgb', if_sam_distillation=False)
    for tag, kw in {"ddd17_train_f2v": dict(split='train', config_option='frame2voxel', separate_pol=False, augmentation=False, skip_ratio=1),
                    "ddd17_valid_f2v": dict(split='valid', config_option='frame2voxel', separate_pol=False, augmentation=False, skip_ratio=1),
                    "ddd17_train_f2v_sep2": dict(split='train', config_option='frame2voxel', separate_pol=True, augmentation=False, skip_ratio=2),
                    "ddd17_train_f2r": dict(split='train', config_option='frame2recon', separate_pol=False, augmentation=False, skip_ratio=1)}.items():
        kw2 = dict(c17, **kw)
        if tag.endswith("sep2"):
            kw2.update(nr_bins_per_data=2, normalize_event=True)
        ds = DDD17Events(root17, **kw2)
        # glob order is filesystem order in the reference (ddd17_events_loader.py:93): compare as a path-keyed set
        out[f"{tag}_len"] = np.array(len(ds))
        order = np.argsort([os.path.relpath(f, root17) for f in ds.files])
        for n, i in enumerate(order[::3]):
            record(out, f"{tag}_{n}", ds[int(i)], root17, 6)
    k = aug_seed(True, True)
    ds = DDD17Events(root17, **dict(c17, split='train', config_option='frame2voxel', separate_pol=False, augmentation=True, skip_ratio=1))
    out["ddd17_aug_pyseed"] = np.array(k)
    random.seed(k)
    torch.manual_seed(99)
    i = int(np.argsort([os.path.relpath(f, root17) for f in ds.files])[2])
    record(out, "ddd17_aug_0", ds[i], root17, 6)
    np.savez_compressed(os.path.join(HERE, "datasets.npz"), **out)
    print("datasets.npz:", len(out), "arrays", os.path.getsize(os.path.join(HERE, "datasets.npz")) // 1024, "KiB")


GROUPS = {"datasets": gen_datasets}
if __name__ == "__main__":
    gen_datasets()
