"""DDD17Events (datasets/ddd17_events_loader.py:34-344) + the memmap helpers of
datasets/extract_data_tools/example_loader_ddd17.py:9-54.

On-disk format per sequence directory: `events.dat.t` int64 [N,1], `events.dat.xyp` int16 [N,3],
`index/index_{10,50,250}ms.npy` rows (timestamp, event_idx, event_idx_before), `segmentation_masks/*.png`.
`__getitem__` returns the reference's 6-tuple (event | frame, label, frame | recon, pl, superpixel, file_path) with the
reference's split rule (`get_split`: dir1 held out), file-name rules, pl / frame / recon / superpixel loading and
augmentation draw order; in the voxel options the first item is `{'events': int64 [N,4] rows (x, y, t, p), 'flip': bool}`:
the 20 chunks are voxelized together on the GPU (`voxelize_batch`), followed by the reference's 346->352 bilinear resize
(align_corners=True, :183-189), the `[:, :-60, :]` crop (:196) and the horizontal flip.  `materialize(idx)` returns the
reference's tuple with the voxel tensor in place."""
import glob
import os
import random
from os.path import basename, dirname, join

import numpy as np
import torch
import torch.nn.functional as f
from torch.utils.data import Dataset

from .. import hip
from . import _io


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


def get_split(dirs, split):
    """ddd17_events_loader.py:19-23: dir1 is the held-out recording."""
    return {"train": [dirs[0], dirs[2], dirs[3], dirs[4], dirs[5]], "valid": [dirs[1]]}[split]


def _sibling(file_path, folder, kind):
    """File-name rules of ddd17_events_loader.py:205-260: `segmentation_masks/segmentation_XXXXXXXX.png` ->
    `<folder>/<kind>_XXXXXXXX.png` in dir0 / dir1 and `<folder>/00XXXXXXXX.png` in the other recordings."""
    path = file_path.replace('segmentation_masks', folder)
    a = path.split('segmentation_')
    if path.split('/')[-3] in ('dir0', 'dir1'):
        path = a[0] + a[1]
        return path.replace(path.split('/')[-1], kind + '_' + path.split('/')[-1])
    return a[0] + '00' + a[1]


class DDD17Events(Dataset):
    def __init__(self, root, split='train', event_representation='voxel_grid', nr_events_data=5, delta_t_per_data=50,
                 nr_bins_per_data=5, require_paired_data=False, separate_pol=False, normalize_event=False, augmentation=False,
                 fixed_duration=False, nr_events_per_data=32000, resize=True, random_crop=False, config_option='',
                 pl_sources='', superpixel_sources='', skip_ratio=1, if_sam_distillation=False):
        data_dirs = sorted(glob.glob(join(root, "dir*")))
        assert len(data_dirs) > 0
        assert split in ["train", "valid", "test"]
        if fixed_duration:
            raise NotImplementedError("fixed_duration windows (np.searchsorted chunking, :160-161) are not used by any shipped config")
        if event_representation != 'voxel_grid':
            raise NotImplementedError("DDD17 configs use event_representation 'voxel_grid'")
        self.root, self.split, self.augmentation = root, split, augmentation
        self.nr_events_data, self.nr_events_per_data = nr_events_data, nr_events_per_data
        self.delta_t_per_data, self.t_interval = delta_t_per_data, -1
        self.nr_events = nr_events_data * nr_events_per_data
        self.nr_temporal_bins, self.separate_pol, self.normalize_event = nr_bins_per_data, separate_pol, normalize_event
        self.event_representation = event_representation
        self.require_paired_data = require_paired_data
        self.shape, self.shape_resize, self.resize = [260, 346], [260, 352], resize
        self.random_crop, self.shape_crop = random_crop, [120, 216]
        self.dirs = get_split(data_dirs, split)
        self.skip_ratio = skip_ratio
        self.files = []
        for d in self.dirs:
            label_files = glob.glob(join(d, "segmentation_masks", "*.png"))      # unsorted, as the reference (:93)
            n = len(label_files)
            if skip_ratio != 1:
                label_files = label_files[:n // skip_ratio + 1]
                print("Seq '{}': '{}' of '{}' data loaded with skipping ratio '{}' .".format(d, len(label_files), n, skip_ratio))
            else:
                print("Seq '{}': '{}' data loaded.".format(d, n))
            self.files += label_files
        self.img_timestamp_event_idx, self.event_data = {}, {}
        self.event_dirs = self.dirs
        for d in self.event_dirs:
            idx, t_ev, xyp_ev, _ = load_files_in_directory(d, self.t_interval)
            self.img_timestamp_event_idx[d] = idx
            self.event_data[d] = [t_ev, xyp_ev]
        self.config_option, self.pl_sources = config_option, pl_sources
        self.superpixel_sources, self.if_sam_distillation = superpixel_sources, if_sam_distillation

    def __len__(self):
        return len(self.files)

    def _resize_label(self, a):
        if self.resize:       # cv2.resize(a, (352, 200), INTER_NEAREST) (:133-137, :238-242, :262-266)
            a = _io.resize_nearest_cv2(a, (self.shape_resize[1], self.shape_resize[0] - 60))
        return a

    def __getitem__(self, idx):
        file_path = self.files[idx]
        label_tensor = torch.from_numpy(np.array(self._resize_label(_io.load_png_gray(file_path)))).long()
        events = None
        if self.config_option in ('recon2voxel', 'frame2voxel'):
            directory = dirname(dirname(file_path))
            img_idx = int(basename(file_path).split("_")[-1].split(".")[0]) - 1
            t_events, xyp_events = self.event_data[directory]
            ev = extract_events_from_memmap(t_events, xyp_events, img_idx, self.img_timestamp_event_idx[directory], False, self.nr_events)
            events = {'events': torch.from_numpy(ev), 'flip': False}
        frame = recon = None
        if self.config_option in ('frame2voxel', 'frame2recon'):
            frame = _io.image_to_chw_float(_sibling(file_path, 'images_aligned', 'img'))
        if self.config_option in ('recon2voxel', 'frame2recon'):
            recon = _io.image_to_chw_float(file_path.replace('segmentation_masks', 'reconstructions'))
        if self.split == 'train':
            pl = torch.tensor(self._resize_label(_io.load_png(_sibling(file_path, self.pl_sources, 'segmentation')))).squeeze(0).long()
        else:
            pl = torch.ones_like(label_tensor)
        if len(self.superpixel_sources) > 1:
            folder = {'sp_slic_rgb': self.superpixel_sources, 'sp_sam_rgb': 'superpixels_sam'}[self.superpixel_sources]
            sp_path = _sibling(file_path, folder, 'img')
            if self.superpixel_sources == 'sp_slic_rgb':
                sp_path = sp_path.replace('.png', '_slic_25.png')
            superpixel = torch.tensor(self._resize_label(_io.load_png(sp_path))).long()
        else:
            superpixel = torch.ones_like(label_tensor)
        opt = self.config_option
        if opt not in ('recon2voxel', 'frame2voxel', 'frame2recon'):
            return None                                                        # the reference falls off the end too
        if self.augmentation:                   # same order of random draws as ddd17_events_loader.py:272-343
            if random.random() >= 0.5:
                if events is not None:
                    events['flip'] = True                                      # torch.flip(event_tensor, [2]) after voxelization
                label_tensor = torch.flip(label_tensor, [1])
                if recon is not None:
                    recon = torch.flip(recon, [2])
                if frame is not None:
                    frame = torch.flip(frame, [2])
                pl = torch.flip(pl, [1])
                superpixel = torch.flip(superpixel, [1])
            if opt == 'frame2recon':
                if random.random() >= 0.5:
                    recon = _io.adjust_brightness(recon, random.uniform(0.8, 1.2))
                    frame = _io.adjust_brightness(frame, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    recon = _io.adjust_contrast(recon, random.uniform(0.8, 1.2))
                    frame = _io.adjust_contrast(frame, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    recon = recon + torch.randn(recon.size()) * 0.05
                    frame = frame + torch.randn(frame.size()) * 0.05
            else:
                x = recon if opt == 'recon2voxel' else frame
                if random.random() >= 0.5:
                    x = _io.adjust_brightness(x, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    x = _io.adjust_contrast(x, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    x = x + torch.randn(x.size()) * 0.05
                if opt == 'recon2voxel':
                    recon = x
                else:
                    frame = x
        if opt == 'recon2voxel':
            return events, label_tensor, recon, pl, superpixel, file_path
        if opt == 'frame2voxel':
            return events, label_tensor, frame, pl, superpixel, file_path
        return frame, label_tensor, recon, pl, superpixel, file_path

    def materialize(self, idx, device='cuda'):
        """The reference's tuple exactly: the voxel tensor (CPU float32) in place of the raw event rows."""
        item = self[idx]
        if isinstance(item[0], dict):
            vox = self.voxelize_batch([item[0]['events']], torch.device(device), flips=[item[0]['flip']])[0].cpu()
            return (vox, *item[1:])
        return item

    def voxelize_batch(self, events_list, device, flips=None):
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
        vox = vox[:, :, :-60, :].contiguous()
        if flips is not None and any(flips):
            idx = torch.tensor([i for i, fl in enumerate(flips) if fl], device=vox.device)
            vox[idx] = torch.flip(vox[idx], [3])
        return vox

    @classmethod
    def build_from_settings(cls, s):
        """Argument mapping of createDDD17EventsDataset (training/base_trainer_ov.py:198-276): the validation set is the
        'valid' split (dir1), never augmented."""
        kw = dict(event_representation=s.event_representation_b, nr_events_data=s.nr_events_data_b,
                  delta_t_per_data=s.delta_t_per_data_b, nr_bins_per_data=s.nr_temporal_bins_b, separate_pol=s.separate_pol_b,
                  normalize_event=s.normalize_event_b, fixed_duration=s.fixed_duration_b, nr_events_per_data=s.nr_events_window_b,
                  config_option=s.config_option, pl_sources=getattr(s, 'pl_sources', ''),
                  superpixel_sources=getattr(s, 'superpixel_sources', ''), skip_ratio=s.skip_ratio,
                  if_sam_distillation=getattr(s, 'if_sam_distillation', False))
        train = cls(s.dataset_path_b, split=s.split_train_b, augmentation=s.data_augmentation_train,
                    require_paired_data=s.require_paired_data_train_b, **kw)
        val = cls(s.dataset_path_b, split='valid', augmentation=False, require_paired_data=s.require_paired_data_val_b, **kw)
        return train, val
