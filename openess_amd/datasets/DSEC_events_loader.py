"""DSECEvents (datasets/DSEC_events_loader.py:6-67): same keyword arguments, returns the train or the val ConcatDataset of
`Sequence`s.  The returned dataset yields RAW event columns in the voxel options and voxelizes whole batches on the GPU
(`dataset.voxelize_batch`, see DSEC/dataset/sequence_ov.py); `build_from_settings` is what BaseTrainer.createDataLoaders
calls with the reference's argument mapping (training/base_trainer_ov.py:93-160,283-305)."""
from pathlib import Path

from ..DSEC.dataset.provider import DatasetProvider


def DSECEvents(dsec_dir, nr_events_data=1, delta_t_per_data=50, nr_events_window=-1, augmentation=False, mode='train',
               task='segmentation', event_representation='voxel_grid', nr_bins_per_data=5, require_paired_data=False,
               separate_pol=False, normalize_event=False, semseg_num_classes=11, fixed_duration=False, resize=False,
               config_option='', pl_sources='', superpixel_sources='', skip_ratio=1, if_sam_distillation=False,
               device_png=False):
    dsec_dir = Path(dsec_dir)
    assert dsec_dir.is_dir()
    provider = DatasetProvider(dsec_dir, mode, event_representation=event_representation, nr_events_data=nr_events_data,
                               delta_t_per_data=delta_t_per_data, nr_events_window=nr_events_window,
                               nr_bins_per_data=nr_bins_per_data, require_paired_data=require_paired_data,
                               normalize_event=normalize_event, separate_pol=separate_pol, semseg_num_classes=semseg_num_classes,
                               augmentation=augmentation, fixed_duration=fixed_duration, resize=resize, config_option=config_option,
                               pl_sources=pl_sources, superpixel_sources=superpixel_sources, skip_ratio=skip_ratio,
                               if_sam_distillation=if_sam_distillation, device_png=device_png)
    return provider.get_train_dataset() if mode == 'train' else provider.get_val_dataset()


def build_from_settings(s):
    kw = dict(dsec_dir=s.dataset_path_b, nr_events_data=s.nr_events_data_b, delta_t_per_data=s.delta_t_per_data_b,
              nr_events_window=s.nr_events_window_b, event_representation=s.event_representation_b,
              nr_bins_per_data=s.nr_temporal_bins_b, separate_pol=s.separate_pol_b, normalize_event=s.normalize_event_b,
              semseg_num_classes=s.semseg_num_classes, fixed_duration=s.fixed_duration_b, config_option=s.config_option,
              pl_sources=getattr(s, 'pl_sources', ''), device_png=getattr(s, 'device_png_decode', False))
    train = DSECEvents(augmentation=s.data_augmentation_train, mode='train', require_paired_data=s.require_paired_data_train_b,
                       superpixel_sources=getattr(s, 'superpixel_sources', ''), skip_ratio=s.skip_ratio,
                       if_sam_distillation=getattr(s, 'if_sam_distillation', False), **kw)
    val = DSECEvents(augmentation=False, mode='val', require_paired_data=s.require_paired_data_val_b, superpixel_sources='',
                     skip_ratio=2, if_sam_distillation=False, **kw)
    return train, val


DSECEvents.build_from_settings = build_from_settings
