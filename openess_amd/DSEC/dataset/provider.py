"""DatasetProvider (DSEC/dataset/provider.py:6-105): the hard-coded train / val sequence lists, ConcatDataset of Sequences,
`.require_paired_data` on the result; the val split always uses skip_ratio 2 and no superpixels (:81-91)."""
from pathlib import Path

import torch

from .sequence_ov import Sequence, voxelize_raw_batch

TRAIN_SEQUENCES = ['zurich_city_00_a', 'zurich_city_01_a', 'zurich_city_02_a', 'zurich_city_04_a', 'zurich_city_05_a',
                   'zurich_city_06_a', 'zurich_city_07_a', 'zurich_city_08_a']
VAL_SEQUENCES = ['zurich_city_13_a', 'zurich_city_14_c', 'zurich_city_15_a']


class DSECConcat(torch.utils.data.ConcatDataset):
    """ConcatDataset + the batched GPU voxelizer: one rectify map per sequence, selected per sub-window by `seg_map`."""
    sensor_hw, crop_rows = (480, 640), 40

    def __getitem__(self, idx):
        item = super().__getitem__(idx)
        if isinstance(item, tuple) and isinstance(item[0], dict):
            import bisect
            item[0]['sequence'] = bisect.bisect_right(self.cumulative_sizes, idx if idx >= 0 else len(self) + idx)
        return item

    def rectify_maps(self, device):
        key = str(device)
        if getattr(self, '_rmaps', {}).get(key) is None:
            import numpy as np
            self._rmaps = getattr(self, '_rmaps', {})
            self._rmaps[key] = torch.from_numpy(np.stack([d.rectify_ev_maps['left'] for d in self.datasets])).to(device)
        return self._rmaps[key]

    def voxelize_batch(self, batch0, device):
        d0 = self.datasets[0]
        nwin = d0.nr_events_data
        seqs = batch0.get('sequence')
        if seqs is not None:
            batch0 = dict(batch0, seg_map=torch.tensor(seqs, dtype=torch.int32).repeat_interleave(nwin))
        return voxelize_raw_batch(batch0, device, self.rectify_maps(device), d0.num_bins, d0.height, d0.width, d0.crop_rows, nwin)


class DatasetProvider:
    def __init__(self, dataset_path, mode='train', event_representation='voxel_grid', nr_events_data=5, delta_t_per_data=20,
                 nr_events_window=-1, nr_bins_per_data=5, require_paired_data=False, normalize_event=False, separate_pol=False,
                 semseg_num_classes=11, augmentation=False, fixed_duration=False, resize=False, config_option='', pl_sources='',
                 superpixel_sources='', skip_ratio=1, if_sam_distillation=False, device_png=False):
        dataset_path = Path(dataset_path)
        train_path, val_path = dataset_path / 'train', dataset_path / 'test'
        assert dataset_path.is_dir(), str(dataset_path)
        assert train_path.is_dir(), str(train_path)
        assert val_path.is_dir(), str(val_path)

        def sequences(path, names, mode_, **kw):
            return [Sequence(child, mode_, event_representation, nr_events_data, delta_t_per_data, nr_events_window, nr_bins_per_data,
                             require_paired_data, normalize_event, separate_pol, semseg_num_classes, augmentation, fixed_duration,
                             resize=resize, config_option=config_option, pl_sources=pl_sources, device_png=device_png, **kw)
                    for child in path.iterdir() if any(k in str(child) for k in names)]
        if mode == 'train':
            self.train_dataset = DSECConcat(sequences(train_path, TRAIN_SEQUENCES, 'train', superpixel_sources=superpixel_sources,
                                                      skip_ratio=skip_ratio, if_sam_distillation=if_sam_distillation))
            self.train_dataset.require_paired_data = require_paired_data
        elif mode == 'val':
            self.val_dataset = DSECConcat(sequences(val_path, VAL_SEQUENCES, 'val', superpixel_sources='', skip_ratio=2,
                                                    if_sam_distillation=False))
            self.val_dataset.require_paired_data = require_paired_data

    def get_train_dataset(self):
        return self.train_dataset

    def get_val_dataset(self):
        return self.val_dataset

    def get_test_dataset(self):
        raise NotImplementedError
