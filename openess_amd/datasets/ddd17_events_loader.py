"""DDD17Events (datasets/ddd17_events_loader.py:34-344) + the memmap helpers of
datasets/extract_data_tools/example_loader_ddd17.py:9-54.

On-disk format per sequence directory: `events.dat.t` int64 [N,1], `events.dat.xyp` int16 [N,3],
`index/index_{10,50,250}ms.npy` rows (timestamp, event_idx, event_idx_before), `segmentation_masks/*.png`.
`__getitem__` returns the reference's tuple, but in the voxel options the first item is the sample's RAW event
rows [N,4] int64 (x, y, t, p): the 20 chunks are voxelized together on the GPU (`voxelize_batch`), followed by the
reference's 346->352 bilinear resize (align_corners=True, :183-189) and the `[:, :-60, :]` crop (:196)."""
import glob
import os
from os.path import basename, dirname, join

import numpy as np
import torch
import torch.nn.functional as f
from torch.utils.data import Dataset

from .. import hip


def load_events(t_file, xyp_file):
    n = int(os.path.getsize(t_file) / 8)
    return (np.memmap(t_file, dtype="int64", mode="r", shape=(n, 1)),
            np.memmap(xyp_file, dtype="int16", mode="r", shape=(n, 3)))


def load_files_in_directory(directory, t_interval=50):
    name = {10: "index_10ms.npy", 50: "index_50ms.npy", 250: "index_250ms.npy"}.get(t_interval, "index_50ms.npy")
    idx = np.load(join(directory, "index", name))
    t_events, xyp_events = load_events(join(directory, "events.dat.t"), join(directory, "events.dat.xyp"))
    masks = sorted(glob.glob(join(directory, "segmentation_masks", "*.png")))
    return idx, t_events, xyp_events, masks


def extract_events_from_memmap(t_events, xyp_events, img_idx, img_timestamp_event_idx, fixed_duration=False, nr_events=32000):
    """Last `nr_events` events before the frame (or the fixed-duration window) as int64 [N,4] (x, y, t, p)."""
    if fixed_duration:
        _, event_idx, event_idx_before = img_timestamp_event_idx[img_idx]
        event_idx_before = max(event_idx_before, 0)
    else:
        _, event_idx, _ = img_timestamp_event_idx[img_idx]
        event_idx_before = max(event_idx - nr_events, 0)
    out = np.empty((event_idx - event_idx_before, 4), dtype=np.int64)
    out[:, 2] = t_events[event_idx_before:event_idx, 0]
    out[:, [0, 1, 3]] = xyp_events[event_idx_before:event_idx]
    return out


class DDD17Events(Dataset):
    def __init__(self, root, split='train', event_representation='voxel_grid', nr_events_data=20, delta_t_per_data=50,
                 nr_bins_per_data=5, require_paired_data=False, separate_pol=False, normalize_event=False, augmentation=False,
                 fixed_duration=False, nr_events_per_data=32000, resize=True, random_crop=False, config_option='frame2voxel',
                 pl_sources='pl_fcclip_rgb', superpixel_sources='sp_sam_rgb', skip_ratio=1, if_sam_distillation=False,
                 dirs=None):
        if augmentation or random_crop or fixed_duration:
            raise NotImplementedError("augmentation / random_crop / fixed_duration are loader-side options not on the hot path")
        self.root, self.split = root, split
        self.nr_events_data, self.nr_events_per_data = nr_events_data, nr_events_per_data
        self.nr_events = nr_events_data * nr_events_per_data
        self.nr_temporal_bins, self.separate_pol, self.normalize_event = nr_bins_per_data, separate_pol, normalize_event
        self.event_representation = event_representation
        self.require_paired_data = require_paired_data
        self.shape, self.shape_resize, self.resize = [260, 346], [260, 352], resize
        self.config_option = config_option
        self.dirs = dirs if dirs is not None else sorted(d for d in glob.glob(join(root, "dir*")) if os.path.isdir(d))
        self.files, self.img_timestamp_event_idx, self.event_data = [], {}, {}
        for d in self.dirs:
            labels = sorted(glob.glob(join(d, "segmentation_masks", "*.png")))
            if skip_ratio != 1:
                labels = labels[:len(labels) // skip_ratio + 1]
            self.files += labels
            idx, t_ev, xyp_ev, _ = load_files_in_directory(d, -1)
            self.img_timestamp_event_idx[d] = idx
            self.event_data[d] = [t_ev, xyp_ev]

    def __len__(self):
        return len(self.files)

    def __getitem__(self, idx):
        from PIL import Image
        mask_file = self.files[idx]
        label = np.array(Image.open(mask_file))
        if self.resize:       # cv2.resize(mask, (352, 200), INTER_NEAREST) in the reference (:133-137)
            label = np.array(Image.fromarray(label).resize((self.shape_resize[1], self.shape_resize[0] - 60), Image.NEAREST))
        label_tensor = torch.from_numpy(label).long()
        directory = dirname(dirname(mask_file))
        img_idx = int(basename(mask_file).split("_")[-1].split(".")[0]) - 1
        t_events, xyp_events = self.event_data[directory]
        events = extract_events_from_memmap(t_events, xyp_events, img_idx, self.img_timestamp_event_idx[directory], False, self.nr_events)
        ones = torch.ones_like(label_tensor)
        return {'events': torch.from_numpy(events)}, label_tensor, torch.zeros(3, *label_tensor.shape), label_tensor, ones, \
            torch.ones(256, 64, 64), mask_file

    def voxelize_batch(self, events_list, device):
        """events_list: per-sample int64 [N_i,4] tensors -> B x (nr_events_data*C) x 200 x 352 float32 on `device`."""
        nwin, C = self.nr_events_data, (2 if self.separate_pol else 1) * self.nr_temporal_bins
        offs, chunks = [0], []
        for ev in events_list:
            n = ev.shape[0] // nwin                  # nr_events_temp = nr_events_loaded // nr_events_data (:152)
            chunks.append(ev[:n * nwin])
            base = offs[-1]
            offs.extend([base + n * (i + 1) for i in range(nwin)])
        ev = torch.cat(chunks).to(device)
        H, W = self.shape
        vox = hip.voxelize_nearest(ev, torch.tensor(offs, dtype=torch.int64), self.nr_temporal_bins, H, W,
                                   separate_pol=self.separate_pol)
        vox = vox.view(len(events_list) * nwin, C, H, W)
        if self.normalize_event:
            vox = torch.stack([hip.masked_normalize(v.contiguous()) for v in vox])
        if self.resize:
            # F.interpolate(bilinear, align_corners=True) 260x346 -> 260x352 (:183-189) on the HIP resampler
            vox = hip.bilinear_resize(vox, size=tuple(self.shape_resize), align_corners=True).contiguous()
        vox = vox.reshape(len(events_list), nwin * C, vox.shape[-2], vox.shape[-1])
        return vox[:, :, :-60, :].contiguous()

    @classmethod
    def build_from_settings(cls, s):
        kw = dict(event_representation=s.event_representation_b, nr_events_data=s.nr_events_data_b,
                  delta_t_per_data=s.delta_t_per_data_b, nr_bins_per_data=s.nr_temporal_bins_b, separate_pol=s.separate_pol_b,
                  normalize_event=s.normalize_event_b, fixed_duration=s.fixed_duration_b, nr_events_per_data=s.nr_events_window_b,
                  config_option=s.config_option, skip_ratio=s.skip_ratio)
        return cls(s.dataset_path_b, split=s.split_train_b, **kw), cls(s.dataset_path_b, split='valid', **kw)
