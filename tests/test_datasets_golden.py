"""Dataset path (SURVEY 8a a2/a5, 8b dataset selector, 8f-2) against golden vectors produced by the REFERENCE's own
`DSECEvents` / `DDD17Events` running over the same deterministic fake trees (tests/golden/gen_golden_datasets.py).

CPU tests: every non-event field bit-exact (crc32 of the tensor bytes), file paths, dataset lengths (sequence lists,
skip ratios, split rule), and -- for the voxel options -- the ORACLE's voxelization of the raw event slices the mirror
selected, bit-exact against the reference's voxel tensor (pins a2: slicing, rectification, float64 promotion, per-chunk time
normalisation, remainder drop, crop).  GPU tests: the batched HIP voxelizer on the same slices, abs 2e-5."""
import os
import random
import zlib

import numpy as np
import pytest
import torch

from tests import synth_datasets as sd
from tests.synth import compact

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMMON = dict(nr_events_data=4, delta_t_per_data=20, nr_events_window=3000, event_representation='voxel_grid', nr_bins_per_data=5,
              require_paired_data=False, separate_pol=False, normalize_event=False, semseg_num_classes=11,
              pl_sources='pl_fcclip_rgb', if_sam_distillation=False)
DSEC_CASES = {
    "dsec_train_f2v": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
    "dsec_train_f2v_slic": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_slic_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
    "dsec_train_f2r": dict(mode='train', config_option='frame2recon', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
    "dsec_train_r2v_fixdur": dict(mode='train', config_option='recon2voxel', superpixel_sources='', augmentation=False, fixed_duration=True, skip_ratio=1),
    "dsec_val_f2v": dict(mode='val', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=1),
    "dsec_train_f2v_skip": dict(mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb', augmentation=False, fixed_duration=False, skip_ratio=2),
}
C17 = dict(event_representation='voxel_grid', nr_events_data=4, delta_t_per_data=50, nr_bins_per_data=5, require_paired_data=False,
           normalize_event=False, fixed_duration=False, nr_events_per_data=700, resize=True, random_crop=False,
           pl_sources='pl_fcclip_rgb', superpixel_sources='sp_sam_rgb', if_sam_distillation=False)
DDD17_CASES = {"ddd17_train_f2v": dict(split='train', config_option='frame2voxel', separate_pol=False, augmentation=False, skip_ratio=1),
               "ddd17_valid_f2v": dict(split='valid', config_option='frame2voxel', separate_pol=False, augmentation=False, skip_ratio=1),
               "ddd17_train_f2v_sep2": dict(split='train', config_option='frame2voxel', separate_pol=True, augmentation=False, skip_ratio=2,
                                            nr_bins_per_data=2, normalize_event=True),
               "ddd17_train_f2r": dict(split='train', config_option='frame2recon', separate_pol=False, augmentation=False, skip_ratio=1)}


@pytest.fixture(scope="module")
def g():
    return dict(np.load(os.path.join(GOLDEN, "datasets.npz")))


@pytest.fixture(scope="module")
def dsec_root(tmp_path_factory):
    return sd.make_dsec_tree(str(tmp_path_factory.mktemp("dsec")))


@pytest.fixture(scope="module")
def ddd17_root(tmp_path_factory):
    return sd.make_ddd17_tree(str(tmp_path_factory.mktemp("ddd17")))


def crc(a):
    a = np.ascontiguousarray(a.numpy() if torch.is_tensor(a) else a)
    return [zlib.crc32(a.tobytes()), a.size]


def oracle_dsec_voxels(ds_seq, ev):
    """Oracle voxelization (numpy, reference type promotion) of the mirror's raw slice: per sub-window VoxelGrid.convert."""
    from oracle import events as oe
    C, H, W = ds_seq.num_bins, ds_seq.height, ds_seq.width
    x, y, t, p = (ev[k].numpy() for k in ('x', 'y', 't', 'p'))
    offs = ev['seg_offsets'].numpy()
    rmap = ds_seq.rectify_ev_maps['left']
    xy = rmap[y.astype(np.int64), x.astype(np.int64)]
    out = np.zeros(((len(offs) - 1) * C, H, W), np.float32)
    for i in range(len(offs) - 1):
        s, e = offs[i], offs[i + 1]
        if e > s:
            out[i * C:(i + 1) * C] = oe.voxelgrid_trilinear(xy[s:e, 0], xy[s:e, 1], p[s:e].astype(np.float32),
                                                            oe.dsec_time_normalise(t[s:e]), C, H, W)
    out = out[:, :H - ds_seq.crop_rows]
    return out[:, :, ::-1].copy() if ev['flip'] else out


def check_item(g, tag, item, root, n_fields, first_override=None, vox_atol=None):
    for j in range(n_fields - 1):
        v = item[j] if not (j == 0 and first_override is not None) else first_override
        if isinstance(v, dict):
            continue
        v = torch.as_tensor(v)
        assert list(v.shape) == list(g[f"{tag}_f{j}_shape"]), (tag, j)
        assert str(v.dtype) == str(g[f"{tag}_f{j}_dtype"]), (tag, j, v.dtype)
        if j == 0 and vox_atol is not None:                       # GPU voxels: tolerance instead of bit-exactness
            sub, ssum, sabs = compact(v.numpy(), n=2048)
            np.testing.assert_allclose(sub, g[f"{tag}_f{j}_sub"], rtol=0, atol=vox_atol)
            assert abs(float(sabs) - float(g[f"{tag}_f{j}_abs"])) <= 1e-5 * float(g[f"{tag}_f{j}_abs"]) + 1e-3
        else:
            assert crc(v) == list(g[f"{tag}_f{j}_crc"]), (tag, j)
    assert os.path.relpath(item[n_fields - 1], root) == str(g[f"{tag}_path"]), tag


def seq_of(ds, idx):
    import bisect
    k = bisect.bisect_right(ds.cumulative_sizes, idx)
    return ds.datasets[k]


@pytest.mark.parametrize("tag", list(DSEC_CASES))
def test_dsec_dataset_matches_reference(g, dsec_root, tag):
    from openess_amd.datasets.DSEC_events_loader import DSECEvents
    ds = DSECEvents(dsec_dir=dsec_root, **COMMON, **DSEC_CASES[tag])
    assert len(ds) == int(g[f"{tag}_len"]) and ds.require_paired_data is False
    # the reference iterates the sequence directories in os-listing order: match items by file path
    by_path = {os.path.relpath(ds[i][-1], dsec_root): i for i in range(len(ds))}
    for n in range(len(ds)):
        i = by_path[str(g[f"{tag}_{n}_path"])]
        item = ds[i]
        first = None
        if isinstance(item[0], dict):
            assert item[0]['x'].dtype == torch.uint16 and item[0]['t'].dtype == torch.int64 and item[0]['p'].dtype == torch.uint8
            first = torch.from_numpy(oracle_dsec_voxels(seq_of(ds, i), item[0]))
        check_item(g, f"{tag}_{n}", item, dsec_root, 7, first_override=first)


def test_dsec_augmentation_and_short_windows(g, dsec_root):
    from openess_amd.datasets.DSEC_events_loader import DSECEvents
    for tag, opt in (("dsec_aug_f2v", "frame2voxel"), ("dsec_aug_f2r", "frame2recon")):
        ds = DSECEvents(dsec_dir=dsec_root, **COMMON, mode='train', config_option=opt, superpixel_sources='sp_sam_rgb',
                        augmentation=True, fixed_duration=False, skip_ratio=1)
        i = {os.path.relpath(ds.datasets[k].label_pathstrings[j], dsec_root): ds.cumulative_sizes[k] - len(ds.datasets[k]) + j
             for k in range(len(ds.datasets)) for j in range(len(ds.datasets[k]))}[str(g[f"{tag}_0_path"])]
        random.seed(int(g[f"{tag}_pyseed"]))
        torch.manual_seed(99)
        item = ds[i]
        first = None
        if isinstance(item[0], dict):
            assert item[0]['flip'] is True
            first = torch.from_numpy(oracle_dsec_voxels(seq_of(ds, i), item[0]))
        check_item(g, f"{tag}_0", item, dsec_root, 7, first_override=first)
    ds = DSECEvents(dsec_dir=dsec_root, **dict(COMMON, nr_events_window=20001, nr_events_data=3), mode='train', config_option='frame2voxel',
                    superpixel_sources='', augmentation=False, fixed_duration=False, skip_ratio=1)
    by_path = {os.path.relpath(ds[i][-1], dsec_root): i for i in range(len(ds))}
    for n in (0, 1):
        i = by_path[str(g[f"dsec_short_{n}_path"])]
        item = ds[i]
        assert item[0]['x'].numel() % 3 == 0                      # remainder of N // nr_events_data dropped (sequence_ov.py:302)
        check_item(g, f"dsec_short_{n}", item, dsec_root, 7, first_override=torch.from_numpy(oracle_dsec_voxels(seq_of(ds, i), item[0])))


def oracle_ddd17_voxels(ds, ev, flip):
    import torch.nn.functional as f
    from oracle import events as oe
    nwin, nb = ds.nr_events_data, ds.nr_temporal_bins
    C = nb * (2 if ds.separate_pol else 1)
    ref = oe.ddd17_event_tensor(ev.numpy(), nwin, (260, 346), nb, ds.separate_pol)
    ref = torch.from_numpy(ref).view(nwin, C, 260, 346)
    if ds.normalize_event:
        ref = torch.stack([torch.from_numpy(oe.masked_normalize(r.numpy())) for r in ref])
    ref = f.interpolate(ref, size=(260, 352), mode='bilinear', align_corners=True).reshape(nwin * C, 260, 352)[:, :-60]
    return torch.flip(ref, [2]) if flip else ref


@pytest.mark.parametrize("tag", list(DDD17_CASES))
def test_ddd17_dataset_matches_reference(g, ddd17_root, tag):
    from openess_amd.datasets.ddd17_events_loader import DDD17Events
    ds = DDD17Events(ddd17_root, **dict(C17, **DDD17_CASES[tag]))
    assert len(ds) == int(g[f"{tag}_len"])
    held_out = os.path.join(ddd17_root, "dir1")
    assert all((f.startswith(held_out)) == (DDD17_CASES[tag]['split'] == 'valid') for f in ds.files)      # get_split: dir1 is validation only
    by_path = {os.path.relpath(f, ddd17_root): i for i, f in enumerate(ds.files)}
    n = 0
    while f"{tag}_{n}_path" in g:
        path = str(g[f"{tag}_{n}_path"])
        if path not in by_path:          # skip_ratio truncates an UNSORTED glob (reference :93-106): membership is filesystem-order dependent
            assert DDD17_CASES[tag]['skip_ratio'] != 1
            n += 1
            continue
        item = ds[by_path[path]]
        first = oracle_ddd17_voxels(ds, item[0]['events'], item[0]['flip']) if isinstance(item[0], dict) else None
        if first is not None and ds.normalize_event:
            # normalize_voxel_grid's mean/std are float32 torch reductions in the reference (data_util.py:38-48): summation order
            sub, _, sabs = compact(first.numpy(), n=2048)
            np.testing.assert_allclose(sub, g[f"{tag}_{n}_f0_sub"], rtol=2e-5, atol=2e-5)
            check_item(g, f"{tag}_{n}", (None,) + tuple(item[1:]), ddd17_root, 6, first_override={})
        else:
            check_item(g, f"{tag}_{n}", item, ddd17_root, 6, first_override=first)
        n += 1
    assert n > 0
    # augmentation: flip + noise branches (python-random seed chosen by the generator so that brightness / contrast are not drawn)
    if tag == "ddd17_train_f2v":
        ds = DDD17Events(ddd17_root, **dict(C17, split='train', config_option='frame2voxel', separate_pol=False, augmentation=True, skip_ratio=1))
        i = {os.path.relpath(f, ddd17_root): i for i, f in enumerate(ds.files)}[str(g["ddd17_aug_0_path"])]
        random.seed(int(g["ddd17_aug_pyseed"]))
        torch.manual_seed(99)
        item = ds[i]
        assert item[0]['flip'] is True
        check_item(g, "ddd17_aug_0", item, ddd17_root, 6, first_override=oracle_ddd17_voxels(ds, item[0]['events'], True))


def test_ddd17_label_resize_cv2_nearest_rule_unpinned():
    """UNPINNED (cv2 is absent in the build image, so no vector generated by the reference's own `cv2.resize(..., INTER_NEAREST)`
    exists for ddd17_events_loader.py:131-136,234-236,260-262): the restatement `_io.resize_nearest_cv2` is checked against
    OpenCV's documented rule src = min(floor(dst * src_size / dst_size), src_size - 1) on hand-computed indices for the sizes the
    loader uses (260 x 346 -> 200 x 352), and against the pixel-centre rule it must NOT be (PIL / torch `nearest`)."""
    from openess_amd.datasets import _io
    src = np.arange(260 * 346, dtype=np.int64).reshape(260, 346)
    out = _io.resize_nearest_cv2(src, (352, 200))
    assert out.shape == (200, 352)
    ys, xs = out // 346, out % 346
    # rows: scale 260 / 200 = 1.3 -> 0, 1, 2, 3, 5, 6, 7, 9, ...; columns: 346 / 352 -> 0, 0, 1, 2, ... (dst 1 -> floor(0.983) = 0)
    assert ys[:8, 0].tolist() == [0, 1, 2, 3, 5, 6, 7, 9] and ys[-1, 0] == 258
    assert xs[0, :5].tolist() == [0, 0, 1, 2, 3] and xs[0, -1] == 345 and xs[0, 59] == 57 and xs[0, 60] == 58
    centre = np.floor((np.arange(352) + 0.5) * 346 / 352).astype(np.int64)      # the rule cv2 does not use
    assert (centre != xs[0]).any()
    rgb = np.stack([src, src + 1, src + 2], -1)
    assert np.array_equal(_io.resize_nearest_cv2(rgb, (352, 200))[..., 1], out + 1)


def test_colour_augmentations_match_torchvision_semantics():
    """torchvision is absent here: adjust_brightness / adjust_contrast are restated (openess_amd/datasets/_io.py) and checked
    against hand-computed values of torchvision's documented blend rule."""
    from openess_amd.datasets import _io
    img = torch.tensor([[[0.2, 0.8]], [[0.4, 0.6]], [[1.0, 0.0]]])
    np.testing.assert_allclose(_io.adjust_brightness(img, 1.5).numpy(), np.clip(img.numpy() * 1.5, 0, 1), atol=1e-7)
    gray = 0.2989 * img[0] + 0.587 * img[1] + 0.114 * img[2]
    exp = np.clip(1.2 * img.numpy() + (1 - 1.2) * float(gray.mean()), 0, 1)
    np.testing.assert_allclose(_io.adjust_contrast(img, 1.2).numpy(), exp, atol=1e-6)


def test_eventslicer_binary_search_equals_linear_scan(dsec_root):
    """get_time_indices_offsets: np.searchsorted == the reference's two linear scans (eventslicer.py:177-203), incl. ties
    and out-of-window times."""
    from openess_amd.DSEC.utils.eventslicer import EventSlicer
    rng = np.random.default_rng(3)
    t = np.sort(rng.integers(0, 50, 200))
    for a, b in [(0, 0), (10, 10), (10, 30), (49, 60), (60, 70), (-5, 3), (25, 25)]:
        i0 = next((i for i in range(t.size) if t[i] >= a), None)
        if t[-1] < a:
            exp = (t.size, t.size)
        else:
            i1 = t.size
            for i in range(t.size - 1, -1, -1):
                if t[i] >= b:
                    i1 = i
                else:
                    break
            exp = (i0, i1)
        assert EventSlicer.get_time_indices_offsets(t, a, b) == exp, (a, b)


# --------------------------------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_dsec_gpu_voxelizer_matches_reference_golden(g, dsec_root):
    """materialize() = the reference's tuple with the voxel tensor built by the batched HIP voxelizer (per-sequence rectify maps,
    fixed-count AND fixed-duration windows, flip)."""
    from openess_amd.datasets.DSEC_events_loader import DSECEvents
    from openess_amd.datasets.synthetic_events import collate
    for tag in ("dsec_train_f2v", "dsec_train_r2v_fixdur", "dsec_val_f2v"):
        ds = DSECEvents(dsec_dir=dsec_root, **COMMON, **DSEC_CASES[tag])
        by_path = {os.path.relpath(ds[i][-1], dsec_root): i for i in range(len(ds))}
        idxs = [by_path[str(g[f"{tag}_{n}_path"])] for n in range(len(ds))]
        batch = collate([ds[i] for i in idxs])                                  # ONE batch spanning both sequences
        vox = ds.voxelize_batch(batch[0], torch.device("cuda")).cpu()
        assert vox.shape == (len(ds), 20, 440, 640)
        for n, i in enumerate(idxs):
            check_item(g, f"{tag}_{n}", ds[i], dsec_root, 7, first_override=vox[n], vox_atol=2e-5)
    ds = DSECEvents(dsec_dir=dsec_root, **COMMON, mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb',
                    augmentation=True, fixed_duration=False, skip_ratio=1)
    i = {os.path.relpath(ds.datasets[k].label_pathstrings[j], dsec_root): ds.cumulative_sizes[k] - len(ds.datasets[k]) + j
         for k in range(len(ds.datasets)) for j in range(len(ds.datasets[k]))}[str(g["dsec_aug_f2v_0_path"])]
    random.seed(int(g["dsec_aug_f2v_pyseed"]))
    torch.manual_seed(99)
    item = collate([ds[i]])
    vox = ds.voxelize_batch(item[0], torch.device("cuda")).cpu()[0]
    sub, _, _ = compact(vox.numpy(), n=2048)
    np.testing.assert_allclose(sub, g["dsec_aug_f2v_0_f0_sub"], rtol=0, atol=2e-5)


@pytest.mark.gpu
def test_ddd17_gpu_voxelizer_matches_reference_golden(g, ddd17_root):
    from openess_amd.datasets.ddd17_events_loader import DDD17Events
    for tag in ("ddd17_train_f2v", "ddd17_valid_f2v"):
        ds = DDD17Events(ddd17_root, **dict(C17, **DDD17_CASES[tag]))
        by_path = {os.path.relpath(f, ddd17_root): i for i, f in enumerate(ds.files)}
        n = 0
        while f"{tag}_{n}_path" in g:
            item = ds.materialize(by_path[str(g[f"{tag}_{n}_path"])])
            check_item(g, f"{tag}_{n}", item, ddd17_root, 6, vox_atol=2e-5)
            n += 1


def _yaml_for(tmp_path, base, dataset_block, **clip):
    import yaml
    cfg = yaml.safe_load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", base)))
    cfg['dataset'] = dataset_block
    cfg['clip'].update(clip)
    cfg['dir']['log'] = str(tmp_path / 'log')
    path = str(tmp_path / 'cfg.yaml')
    yaml.safe_dump(cfg, open(path, 'w'))
    return path


@pytest.mark.gpu
@pytest.mark.parametrize("dataset", ["DSEC", "DDD17"])
def test_train_py_on_real_dataset_layout(dataset, dsec_root, ddd17_root, tmp_path):
    """train.py -> OpenESSPretrainModel -> DSECEvents / DDD17Events over an on-disk tree in the reference's layout -> pinned raw
    event batches -> GPU voxelizer -> one epoch of frame2voxel pre-training (+ superpixel InfoNCE) -> validation -> checkpoint:
    the reference's config schema drives the real dataset path end to end (SURVEY 8b 'drops into config/ pipelines')."""
    import train
    from openess_amd.config.settings import Settings
    common = dict(nr_events_data=2, nr_events_files_per_data=None, fixed_duration=False, delta_t_per_data=50,
                  require_paired_data_train=False, require_paired_data_val=False, event_representation='voxel_grid', nr_temporal_bins=5,
                  separate_pol=False, normalize_event=False)
    if dataset == "DSEC":
        block = {'name_b': 'DSEC_events', 'DSEC_events': dict(common, dataset_path=dsec_root, shape=[440, 640], nr_events_window=3000)}
        clip = dict(superpixel_size=100)
    else:
        block = {'name_b': 'DDD17_events', 'DDD17_events': dict(common, dataset_path=ddd17_root, split_train='train', shape=[200, 352],
                                                              nr_events_window=700)}
        clip = dict(superpixel_size=25)
    path = _yaml_for(tmp_path, "pretrain_dsec_synthetic.yaml", block, **clip)
    if dataset == "DDD17":
        import yaml
        cfg = yaml.safe_load(open(path))
        cfg['task']['semseg_num_classes'] = 6
        yaml.safe_dump(cfg, open(path, 'w'))
    train.seed_everything()
    s = Settings(path, generate_log=False)
    s.ckpt_dir = str(tmp_path)
    trainer, which = train.build_trainer(s)
    assert which == 'pretraining'
    n_train = len(trainer.train_loader_sensor_b.dataset)
    assert n_train == (6 if dataset == "DSEC" else 20)              # 2 train sequences x 3 frames / 5 train dirs x 4 frames
    trainer.pretraining()
    assert trainer.step_count == n_train // 2 and 'Epoch_0.pt' in os.listdir(str(tmp_path))
    trainer.valEpochs()
    assert 0.0 <= float(trainer.last_val_metrics['miou']) <= 100.0


@pytest.mark.gpu
def test_dsec_device_png_decode_equals_host_path(dsec_root):
    """with device_png the loader ships the label / pseudo-label / superpixel PNG files undecoded and
    BaseTrainer.prepare_batch decodes the batch on the GPU (flips included): the tensors the step receives are bit-identical to
    the host path's (PIL in the loader = the reference), with and without augmentation."""
    import torch
    from openess_amd import hip
    from openess_amd.datasets.DSEC_events_loader import DSECEvents
    from openess_amd.datasets.synthetic_events import collate
    for aug in (False, True):
        kw = dict(dsec_dir=dsec_root, **COMMON, mode='train', config_option='frame2voxel', superpixel_sources='sp_sam_rgb',
                  augmentation=aug, fixed_duration=False, skip_ratio=1)
        host, dev = DSECEvents(**kw), DSECEvents(**kw, device_png=True)
        assert len(host) == len(dev)
        idx = list(range(min(len(host), 6)))
        items = []
        for ds in (host, dev):
            random.seed(5)
            torch.manual_seed(5)
            items.append(collate([ds[i] for i in idx]))
        flipped = 0
        for slot in (1, 3, 4):                                    # label, pseudo-label, superpixel
            want, got = items[0][slot], items[1][slot]
            assert isinstance(got, dict) and 'png_bytes' in got
            maps, st = hip.png_decode_gray8_batch(got['png_bytes'].cuda(), got['png_lengths'], got['hw'][0], got['hw'][1], got['flip'])
            assert int(st.abs().sum()) == 0
            assert torch.equal(maps.cpu(), want)
            flipped += sum(got['flip'])
        assert (flipped > 0) == aug or not aug
