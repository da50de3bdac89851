"""Plugin surface: Settings attribute names, every reference YAML loads (when the reference tree is
present), trainer dispatch order, checkpoint format, DDD17 on-disk reader."""
import glob
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = os.path.join(HERE, "configs")
REF_CFG = "/root/reference/config"


def test_settings_attributes():
    from openess_amd.config.settings import Settings
    s = Settings(os.path.join(CFG, "pretrain_dsec_synthetic.yaml"), generate_log=False)
    assert s.dataset_name_b == 'DSEC_events' and s.sensor_b_name == 'events'
    assert s.input_channels_b == 5 and s.nr_events_data_b == 3 and s.img_size_b == [64, 96]
    assert s.semseg_num_classes == 11 and s.semseg_ignore_label == 255 and len(s.semseg_class_names) == 11
    assert s.semseg_color_map.shape == (11, 3) and tuple(s.semseg_color_map[10]) == (255, 0, 0)
    assert s.config_option == 'frame2voxel' and s.if_pretraining and s.if_spatial_contrastive and s.superpixel_size == 25
    assert s.batch_size_b == 2 and s.lr_voxel == 5e-4 and s.task_loss == ['dice', 'cross_entropy']
    assert s.if_linear_probing is False and s.use_amp is False and s.frozen_backbone is False
    assert s.e2vid_config.no_normalize is False and s.e2vid_config.no_recurrent is False
    assert s.synthetic_data


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="reference tree not present (GPU box)")
def test_every_reference_yaml_loads_and_dispatches(monkeypatch):
    """(v) of SURVEY.md section 4: YAML round trip of every file under the reference's config/."""
    from openess_amd.config.settings import Settings
    monkeypatch.setenv("OPENESS_ALLOW_MISSING_DATA", "1")
    files = sorted(glob.glob(os.path.join(REF_CFG, "**", "*.yaml"), recursive=True))
    assert len(files) >= 30
    kinds = set()
    for fpath in files:
        s = Settings(fpath, generate_log=False)
        assert s.config_option in ('recon2voxel', 'frame2voxel', 'frame2recon')
        kind = ('sup' if s.if_supervised_only else 'pre' if getattr(s, 'if_pretraining', False)
                else 'ft' if getattr(s, 'if_finetuning', False) else 'lp' if s.if_linear_probing else 'openess')
        kinds.add(kind)
        if '/linear_probe/' in fpath:
            # quirk reproduced: top-level if_linear_probing is ignored -> these files dispatch to OpenESSModel
            assert kind in ('openess', 'ft', 'pre', 'sup')
    assert {'pre', 'ft'} <= kinds


def test_reference_trainer_module_paths_exist():
    """train.py:4-8 of the reference imports its five trainers from five modules; the import switch of INTEGRATION.md needs
    every one of those module paths and class names here (import only: constructing a trainer needs the GPU)."""
    import importlib
    pairs = [("sup_only_trainer", "SupOnlyModel"), ("pretrain_trainer", "OpenESSPretrainModel"),
             ("finetune_trainer", "OpenESSFineTuneModel"), ("linear_probe_trainer", "OpenESSLinearProbeModel"),
             ("openess_trainer", "OpenESSModel")]
    from openess_amd.training.base_trainer_ov import BaseTrainer
    classes = []
    for mod, cls in pairs:
        m = importlib.import_module(f"openess_amd.training.{mod}")
        c = getattr(m, cls)
        assert issubclass(c, BaseTrainer) and c.__module__ == m.__name__, (mod, cls)       # defined there, not re-exported
        classes.append(c)
    assert len(set(classes)) == 5
    # the three stage-2/3 trainers differ exactly where the reference's files differ (constructor flags, AMP branch)
    from types import SimpleNamespace
    s = SimpleNamespace(if_linear_probing=True, if_finetuning=True, frozen_backbone=True, use_amp=True)
    mk = lambda c: c.__new__(c)
    ft, lp, so = mk(classes[2]), mk(classes[3]), mk(classes[0])
    for t in (ft, lp, so):
        t.settings = s
    assert ft.deeplab_kwargs() == {'if_finetuning': True, 'frozen_backbone': True} and ft.backend_kwargs() == {}
    assert lp.deeplab_kwargs() == {'if_linear_probing': True} and lp.backend_kwargs() == {'if_linear_probing': True}
    assert so.deeplab_kwargs() == {} and so.backend_kwargs() == {} and so.amp_requested() and not ft.amp_requested()


def test_png_status_queue_raises_with_slot_and_file_name():
    """device_png: per-slot decode status vectors are queued and checked off the hot path; a non-zero status raises and names the
    slot, the sample and the reason; an all-zero queue is consumed silently; below the check interval nothing is read."""
    from types import SimpleNamespace
    from openess_amd.training.base_trainer_ov import BaseTrainer
    t = BaseTrainer.__new__(BaseTrainer)
    t.settings = SimpleNamespace(png_check_every=2)
    ok = torch.zeros(2, dtype=torch.int32)
    t._png_pending = [(2, ok, ['a.png', 'b.png']), (4, ok, None)]
    t._png_batches = 1                                            # both slots came from ONE batch
    t.check_png_status()                                          # 1 batch < png_check_every = 2: not checked yet
    assert len(t._png_pending) == 2
    t._png_batches = 2                                            # the interval counts batches, not PNG slots
    t.check_png_status()
    assert t._png_pending == [] and t._png_batches == 0
    t._png_pending, t._png_batches = [(2, ok, None)], 1
    t.check_png_status(force=True)
    assert t._png_pending == []
    t._png_pending = [(2, ok, ['a.png', 'b.png']), (5, torch.tensor([0, 3], dtype=torch.int32), ['c.png', 'd.png'])]
    t._png_batches = 1
    with pytest.raises(RuntimeError, match=r"batch slot 5, sample d\.png: unsupported"):
        t.check_png_status(force=True)
    assert t._png_pending == []


def test_checkpoint_format_roundtrip(tmp_path):
    from openess_amd.utils.saver import CheckpointSaver
    models = {'back_end': torch.nn.Linear(3, 2), 'model_frame': torch.nn.Linear(2, 2), 'model_recon': torch.nn.Linear(4, 1)}
    sv = CheckpointSaver(str(tmp_path))
    path = sv.save_checkpoint_model(models, 3, 17)
    assert os.path.basename(path) == 'Epoch_3.pt'
    ck = torch.load(path)
    assert set(ck) == {'back_end', 'model_recon', 'epoch', 'step_count'}        # model_frame is not saved (saver.py:31-42)
    tgt = {'back_end': torch.nn.Linear(3, 2), 'model_recon': torch.nn.Linear(5, 1)}      # shape mismatch is filtered
    before = tgt['model_recon'].weight.clone()
    sv.load_pretrained_weights(tgt, tgt.keys(), path)
    assert torch.equal(tgt['back_end'].weight, models['back_end'].weight)
    assert torch.equal(tgt['model_recon'].weight, before)
    assert os.path.basename(sv.save_checkpoint_model_single(models, 0, 0)) == 'ckp.pt'
    # front_sensor_b / e2vid_decoder are never taken from a stage-1 checkpoint (utils/saver.py:78-79)
    torch.save({'front_sensor_b': torch.nn.Linear(2, 2).state_dict(), 'back_end': models['back_end'].state_dict()}, str(tmp_path / 'x.pt'))
    tgt = {'front_sensor_b': torch.nn.Linear(2, 2), 'back_end': torch.nn.Linear(3, 2)}
    keep = tgt['front_sensor_b'].weight.clone()
    sv.load_pretrained_weights(tgt, tgt.keys(), str(tmp_path / 'x.pt'))
    assert torch.equal(tgt['front_sensor_b'].weight, keep) and torch.equal(tgt['back_end'].weight, models['back_end'].weight)
    with pytest.raises(KeyError):
        sv.load_checkpoint(tgt, {}, checkpoint_file=str(tmp_path / 'x.pt'))


def test_ddd17_memmap_reader(tmp_path):
    """Raw event rows of a sample == the memmap slice the reference extracts (example_loader_ddd17.py:39-54)."""
    from openess_amd.datasets.ddd17_events_loader import DDD17Events
    from tests import synth_datasets as sd
    root = sd.make_ddd17_tree(str(tmp_path))
    ds = DDD17Events(root, nr_events_data=4, nr_events_per_data=500, nr_bins_per_data=5, config_option='frame2voxel',
                     pl_sources='pl_fcclip_rgb', superpixel_sources='sp_sam_rgb')
    assert len(ds) == 5 * 4 and not any('dir1' in f for f in ds.files)             # get_split('train'): dir1 held out
    i = next(k for k, f in enumerate(ds.files) if f.endswith('dir2/segmentation_masks/segmentation_00000002.png'))
    item = ds[i]
    ev = item[0]['events'].numpy()
    d = os.path.join(root, 'dir2')
    idx = np.load(os.path.join(d, 'index', 'index_50ms.npy'))
    t = np.fromfile(os.path.join(d, 'events.dat.t'), dtype=np.int64)
    xyp = np.fromfile(os.path.join(d, 'events.dat.xyp'), dtype=np.int16).reshape(-1, 3)
    hi = idx[1, 1]
    lo = max(hi - 4 * 500, 0)
    assert ev.shape == (hi - lo, 4) and ev.dtype == np.int64
    assert np.array_equal(ev[:, 2], t[lo:hi]) and np.array_equal(ev[:, [0, 1, 3]], xyp[lo:hi].astype(np.int64))
    assert item[1].shape == (200, 352) and item[1].dtype == torch.int64
    assert len(item) == 6 and item[2].shape == (3, 200, 352) and item[3].shape == (200, 352)      # frame, pl (pseudo-labels, not GT)
    assert not torch.equal(item[3], item[1])


@pytest.mark.gpu
def test_ddd17_batch_voxelization_matches_oracle(tmp_path):
    import torch.nn.functional as f
    from openess_amd.datasets.ddd17_events_loader import DDD17Events
    from oracle import events as oe
    from tests import synth_datasets as sd
    root = sd.make_ddd17_tree(str(tmp_path))
    ds = DDD17Events(root, nr_events_data=4, nr_events_per_data=500, nr_bins_per_data=5, config_option='frame2voxel',
                     pl_sources='pl_fcclip_rgb', superpixel_sources='sp_sam_rgb')
    evs = [ds[i][0]['events'] for i in (1, 2)]
    vox = ds.voxelize_batch(evs, torch.device("cuda"))
    assert vox.shape == (2, 20, 200, 352)
    for b, ev in enumerate(evs):
        ref = oe.ddd17_event_tensor(ev.numpy(), 4, (260, 346), 5, False)           # 20 x 260 x 346
        ref = f.interpolate(torch.from_numpy(ref).view(4, 5, 260, 346), size=(260, 352), mode='bilinear', align_corners=True)
        ref = ref.reshape(20, 260, 352)[:, :-60]
        np.testing.assert_allclose(vox[b].cpu().numpy(), ref.numpy(), rtol=0, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,loop", [("pretrain_dsec_synthetic.yaml", "pretraining"), ("finetune_dsec_synthetic.yaml", "training"),
                                      ("openess_dsec_synthetic.yaml", "training")])
def test_train_py_dispatch_and_one_epoch(cfg, loop, tmp_path, monkeypatch):
    """train.py's dispatch + one synthetic epoch end to end (voxelizer -> step -> checkpoint in the reference format)."""
    import train
    from openess_amd.config.settings import Settings
    train.seed_everything()
    s = Settings(os.path.join(CFG, cfg), generate_log=False)
    s.ckpt_dir = str(tmp_path)
    trainer, which = train.build_trainer(s)
    assert which == loop
    if cfg.startswith("openess"):
        assert type(trainer).__name__ == "OpenESSModel"
    getattr(trainer, which)()
    assert trainer.epoch_count == 1 and trainer.step_count == len(trainer.train_loader_sensor_b)
    saved = os.listdir(str(tmp_path))
    assert saved == (['Epoch_0.pt'] if loop == 'pretraining' else ['ckp.pt'])
    if loop == 'training':
        assert 0.0 <= float(trainer.last_val_metrics['miou']) <= 100.0


@pytest.mark.gpu
def test_ingest_prefetch_equals_in_order_loop(tmp_path):
    """BaseTrainer.device_batches (side-stream H2D + voxelizer of batch i+1 under step i, two loader workers, pinned memory)
    against the in-order path: same batches in the same order -> bit-identical weights after the epoch (every kernel of the
    pixel-distillation step is bit-repeatable)."""
    import torch
    import train
    from openess_amd.config.settings import Settings
    finals = []
    for prefetch in (True, False):
        train.seed_everything()
        s = Settings(os.path.join(CFG, "pretrain_dsec_synthetic.yaml"), generate_log=False)
        s.ckpt_dir = str(tmp_path)
        s.if_spatial_contrastive = False
        s.synthetic_length, s.num_cpu_workers, s.ingest_prefetch, s.save_checkpoint = 10, 2, prefetch, False
        trainer, _ = train.build_trainer(s)
        torch.manual_seed(77)                       # the DataLoader's shuffle draws its base seed here
        trainer.trainEpoch()
        assert trainer.step_count == 5
        torch.cuda.synchronize()
        finals.append({k + "." + n: p.detach().clone() for k, m in trainer.models_dict.items() for n, p in m.named_parameters()
                       if p.requires_grad})
    assert any(not torch.equal(v, finals[1][k]) for k, v in finals[0].items()) is False


def test_trainer_dataset_factories_equal_the_loaders_own_mapping(tmp_path):
    """BaseTrainer.createDSECDataset / createDDD17EventsDataset (base_trainer_ov.py:93-183, 187-276 of the reference, positional
    signatures) must build the same (train, validation) datasets as the loaders' settings mapping: same samples, same flags."""
    from types import SimpleNamespace
    from openess_amd.datasets.DSEC_events_loader import DSECEvents
    from openess_amd.datasets.ddd17_events_loader import DDD17Events
    from openess_amd.training.base_trainer_ov import BaseTrainer
    from tests import synth_datasets as sd
    t = BaseTrainer.__new__(BaseTrainer)
    dsec = sd.make_dsec_tree(str(tmp_path / "dsec"))
    s = SimpleNamespace(dataset_name_b='DSEC_events', dataset_path_b=dsec, batch_size_b=2, nr_events_data_b=3, delta_t_per_data_b=50,
                        nr_events_window_b=200, data_augmentation_train=True, event_representation_b='voxel_grid', nr_temporal_bins_b=5,
                        require_paired_data_train_b=False, require_paired_data_val_b=True, separate_pol_b=False, normalize_event_b=False,
                        semseg_num_classes=11, fixed_duration_b=False, config_option='frame2voxel', pl_sources='pl_fcclip_rgb',
                        superpixel_sources='sp_sam_rgb', skip_ratio=1, if_sam_distillation=False, device_png_decode=False)
    t.settings = s
    tr, va = t.createDSECDataset(s.dataset_name_b, s.dataset_path_b, s.batch_size_b, s.nr_events_data_b, s.delta_t_per_data_b,
                                 s.nr_events_window_b, s.data_augmentation_train, s.event_representation_b, s.nr_temporal_bins_b,
                                 s.require_paired_data_train_b, s.require_paired_data_val_b, s.separate_pol_b, s.normalize_event_b,
                                 s.semseg_num_classes, s.fixed_duration_b, s.config_option, s.pl_sources, s.superpixel_sources,
                                 s.skip_ratio, s.if_sam_distillation)
    tr2, va2 = DSECEvents.build_from_settings(s)
    assert len(tr) == len(tr2) > 0 and len(va) == len(va2) > 0
    assert getattr(va, 'require_paired_data', None) == getattr(va2, 'require_paired_data', None)
    a, b = va[0], va2[0]
    assert len(a) == len(b) and a[-1] == b[-1] and torch.equal(a[1], b[1])
    ddd = sd.make_ddd17_tree(str(tmp_path / "ddd17"))
    s2 = SimpleNamespace(dataset_name_b='DDD17_events', dataset_path_b=ddd, split_train_b='train', batch_size_b=2, nr_events_data_b=4,
                         delta_t_per_data_b=50, nr_events_window_b=500, data_augmentation_train=False, event_representation_b='voxel_grid',
                         nr_temporal_bins_b=5, require_paired_data_train_b=False, require_paired_data_val_b=True, separate_pol_b=False,
                         normalize_event_b=False, fixed_duration_b=False, config_option='frame2voxel', pl_sources='pl_fcclip_rgb',
                         superpixel_sources='sp_sam_rgb', skip_ratio=1, if_sam_distillation=False)
    t.settings = s2
    tr, va = t.createDDD17EventsDataset(s2.dataset_name_b, s2.dataset_path_b, s2.split_train_b, s2.batch_size_b, s2.nr_events_data_b,
                                        s2.delta_t_per_data_b, s2.nr_events_window_b, s2.data_augmentation_train,
                                        s2.event_representation_b, s2.nr_temporal_bins_b, s2.require_paired_data_train_b,
                                        s2.require_paired_data_val_b, s2.separate_pol_b, s2.normalize_event_b, s2.fixed_duration_b,
                                        s2.config_option, s2.pl_sources, s2.superpixel_sources, s2.skip_ratio, s2.if_sam_distillation)
    tr2, va2 = DDD17Events.build_from_settings(s2)
    assert len(tr) == len(tr2) > 0 and len(va) == len(va2) > 0
    a, b = tr[1], tr2[1]
    assert len(a) == len(b) and torch.equal(a[1], b[1]) and torch.equal(a[0]['events'], b[0]['events'])
