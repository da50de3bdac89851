"""Deterministic FAKE on-disk datasets in the reference's directory layouts (DSEC-Semantic, DDD17-Seg), written into a
temporary directory by the golden generator (tests/golden/gen_golden_datasets.py, which then runs the REFERENCE's own
`DSECEvents` / `DDD17Events` over them) and again by the tests (which run this repo's mirrors over them).

DSEC event / rectify-map containers: the real dataset ships HDF5 (`events/left/events.h5`, `rectify_map.h5`).  h5py is not
installed in this image, so the fake tree uses the UNPACKED layout that `openess_amd.DSEC.utils.eventslicer` also reads
(one .npy per HDF5 dataset in a sibling `<name>_h5/` directory; `tools/dsec_unpack_h5.py` produces it from real files);
the golden generator hands the reference an `h5py.File` look-alike over the same .npy files."""
import os

import numpy as np

from openess_amd.datasets._synth import rectify_map

DSEC_TRAIN_SEQS = ("zurich_city_00_a", "zurich_city_05_a")
DSEC_VAL_SEQS = ("zurich_city_13_a",)
H_SENSOR, W_SENSOR = 480, 640


def _png(path, arr):
    from PIL import Image
    os.makedirs(os.path.dirname(path), exist_ok=True)
    Image.fromarray(arr).save(path)


def write_unpacked_h5(h5_path, datasets):
    """`datasets`: {'events/p': array, ...} -> <h5_path minus '.h5'>_h5/<name with '/' -> '_'>.npy ; also touches h5_path."""
    d = h5_path[:-3] + "_h5"
    os.makedirs(d, exist_ok=True)
    for name, arr in datasets.items():
        np.save(os.path.join(d, name.replace("/", "_") + ".npy"), np.asarray(arr))
    open(h5_path, "ab").close()            # the reference asserts nothing about it, but keep the canonical file name present


def make_dsec_sequence(seq_dir, seed, n_labels=9, n_events=60000, span_ms=900, t_offset=1_000_000, ragged_tail=True):
    """One DSEC sequence: `n_labels` label frames (the reference drops the first 6 timestamps and then 6 more frames,
    sequence_ov.py:93,116-118), events over `span_ms` ms."""
    rng = np.random.default_rng(seed)
    n_ts = n_labels + 6
    # label timestamps (absolute us) spread over the second half of the recording, deliberately NOT multiples of 1000
    ts = t_offset + np.sort(rng.choice(np.arange(span_ms * 400, span_ms * 990), n_ts, replace=False)).astype(np.int64) + 137
    os.makedirs(os.path.join(seq_dir, "semantic"), exist_ok=True)
    np.savetxt(os.path.join(seq_dir, "semantic", "semantic_timestamps.txt"), ts, fmt="%d")
    t = np.sort(rng.integers(0, span_ms * 1000, n_events)).astype(np.int64)          # relative to t_offset, ties allowed
    x = rng.integers(0, W_SENSOR, n_events).astype(np.uint16)
    y = rng.integers(0, H_SENSOR, n_events).astype(np.uint16)
    p = rng.integers(0, 2, n_events).astype(np.uint8)
    ms_to_idx = np.searchsorted(t, np.arange(span_ms + 1) * 1000, side="left").astype(np.uint64)
    write_unpacked_h5(os.path.join(seq_dir, "events", "left", "events.h5"),
                      {"events/p": p, "events/x": x, "events/y": y, "events/t": t, "ms_to_idx": ms_to_idx,
                       "t_offset": np.int64(t_offset)})
    write_unpacked_h5(os.path.join(seq_dir, "events", "left", "rectify_map.h5"),
                      {"rectify_map": rectify_map(H_SENSOR, W_SENSOR, seed=seed % 5 + 1)})
    Hn = H_SENSOR - 40
    for i in range(n_labels):
        name = f"{i:06d}.png"
        lab = rng.integers(0, 11, (Hn // 8, W_SENSOR // 8)).astype(np.uint8).repeat(8, 0).repeat(8, 1)
        lab[rng.uniform(size=lab.shape) < 0.03] = 255
        _png(os.path.join(seq_dir, "semantic", "left", "11classes", name), lab)
        _png(os.path.join(seq_dir, "images_aligned", "left", name), rng.integers(0, 256, (Hn, W_SENSOR, 3)).astype(np.uint8))
        _png(os.path.join(seq_dir, "reconstructions", "left", name), rng.integers(0, 256, (Hn, W_SENSOR, 3)).astype(np.uint8))
        _png(os.path.join(seq_dir, "pl_fcclip_rgb", "left", name),
             rng.integers(0, 11, (Hn // 4, W_SENSOR // 4)).astype(np.uint8).repeat(4, 0).repeat(4, 1))
        sp = rng.integers(0, 140, (Hn // 20, W_SENSOR // 20)).astype(np.uint8).repeat(20, 0).repeat(20, 1)   # ids may exceed superpixel_size
        _png(os.path.join(seq_dir, "sp_sam_rgb", "left", name), sp)
        _png(os.path.join(seq_dir, "sp_slic_rgb", "left", name.replace(".png", "_slic_100.png")), (sp // 2).astype(np.uint8))


def make_dsec_tree(root):
    for k, name in enumerate(DSEC_TRAIN_SEQS):
        make_dsec_sequence(os.path.join(root, "train", name), seed=100 + k)
    # a directory that the provider must skip (not in the hard-coded name lists, provider.py:36-40)
    os.makedirs(os.path.join(root, "train", "zurich_city_03_a"), exist_ok=True)
    for k, name in enumerate(DSEC_VAL_SEQS):
        make_dsec_sequence(os.path.join(root, "test", name), seed=200 + k, n_labels=11)
    return root


def make_ddd17_tree(root, n_dirs=6, n_frames=4, n_events=24000):
    """dir0..dir5 (get_split needs six, ddd17_events_loader.py:19-23) with events.dat.{t,xyp}, index/index_*.npy,
    segmentation_masks / images_aligned / reconstructions / pl / superpixel PNGs under the reference's naming rules
    (:205-260: dir0/dir1 use `img_XXXXXXXX.png` / `segmentation_XXXXXXXX.png`, the others prefix the number with '00')."""
    for d in range(n_dirs):
        rng = np.random.default_rng(500 + d)
        dd = os.path.join(root, f"dir{d}")
        os.makedirs(os.path.join(dd, "index"), exist_ok=True)
        t = np.sort(rng.integers(0, 2 * 10 ** 6, n_events)).astype(np.int64).reshape(-1, 1)
        xyp = np.stack([rng.integers(0, 346, n_events), rng.integers(0, 260, n_events), rng.integers(0, 2, n_events)], -1).astype(np.int16)
        t.tofile(os.path.join(dd, "events.dat.t"))
        xyp.tofile(os.path.join(dd, "events.dat.xyp"))
        ends = np.sort(rng.choice(np.arange(n_events // 3, n_events), n_frames, replace=False))
        idx = np.stack([t[ends - 1, 0], ends, np.maximum(ends - 3000, 0)], -1).astype(np.int64)
        for nm in ("index_10ms.npy", "index_50ms.npy", "index_250ms.npy"):
            np.save(os.path.join(dd, "index", nm), idx)
        for i in range(n_frames):
            num = f"{i + 1:08d}"
            _png(os.path.join(dd, "segmentation_masks", f"segmentation_{num}.png"),
                 rng.integers(0, 6, (25, 43)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:200, :346])
            if d in (0, 1):
                img_name, pl_name = f"img_{num}.png", f"segmentation_{num}.png"
            else:
                img_name = pl_name = f"00{num}.png"
            _png(os.path.join(dd, "images_aligned", img_name), rng.integers(0, 256, (200, 352, 3)).astype(np.uint8))
            _png(os.path.join(dd, "reconstructions", f"segmentation_{num}.png"), rng.integers(0, 256, (200, 352, 3)).astype(np.uint8))
            _png(os.path.join(dd, "pl_fcclip_rgb", pl_name),
                 rng.integers(0, 6, (25, 43)).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:200, :346])
            _png(os.path.join(dd, "superpixels_sam", img_name),
                 rng.integers(0, 40, (20, 35)).astype(np.uint8).repeat(10, 0).repeat(10, 1)[:200, :346])
    return root
