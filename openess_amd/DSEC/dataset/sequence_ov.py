"""Sequence (DSEC/dataset/sequence_ov.py:24-463): one DSEC-Semantic recording as a torch Dataset.  Same constructor
arguments, same file-name rules, same order of `random` draws in the augmentation, same returned tuple
    (event | frame, label, frame | recon, pl, superpixel, sam_feat, file_path)
with ONE difference that is the point of the MI355X path: in the voxel options the first item is not a pre-built
100x440x640 voxel tensor (the reference builds it with 8 joblib threads in the loader worker, :281-307, and ships
901 MB per batch over a pageable H2D copy) but the sample's RAW event columns

    {'x': uint16[N], 'y': uint16[N], 't': int64[N], 'p': uint8[N], 'seg_offsets': int64[nr_events_data+1], 'flip': bool}

(13 bytes per event; `DataLoader(pin_memory=True)` pins them).  `voxelize_batch` turns a collated batch of these into the
reference's tensor on the GPU: rectification gather, per-sub-window time normalisation, tri-linear splat, the
`[:, :-40, :]` crop and the horizontal flip, all in `oess_voxelize_dsec_raw` (+ one flip kernel).  `materialize(index)`
returns the reference's tuple exactly (voxel tensor first) for drop-in use and for the golden tests."""
import os
import random
from pathlib import Path

import numpy as np
import torch
from torch.utils.data import Dataset

from ...datasets import _io
from ..utils.eventslicer import EventSlicer, open_h5


class Sequence(Dataset):
    def __init__(self, seq_path, mode='train', event_representation='voxel_grid', nr_events_data=5, delta_t_per_data=20,
                 nr_events_per_data=100000, nr_bins_per_data=5, require_paired_data=False, normalize_event=False,
                 separate_pol=False, semseg_num_classes=11, augmentation=False, fixed_duration=False, remove_time_window=250,
                 resize=False, config_option='', pl_sources='', superpixel_sources='', skip_ratio=1, if_sam_distillation=False,
                 device_png=False):
        # device_png (extension): label / pseudo-label / superpixel maps leave __getitem__ as the PNG FILE BYTES and are
        # decoded for the whole batch on the GPU (hip.png_decode_gray8_batch in BaseTrainer.prepare_batch), flips included
        self.device_png = bool(device_png)
        seq_path = Path(seq_path)
        assert nr_bins_per_data >= 1
        assert seq_path.is_dir()
        if resize:
            raise NotImplementedError("resize=True (448x640, cv2) is not used by any shipped config (sequence_ov.py:58-62)")
        if event_representation != 'voxel_grid' or normalize_event:
            raise NotImplementedError("DSEC configs use event_representation 'voxel_grid' with normalize_event False")
        self.sequence_name, self.mode, self.skip_ratio = seq_path.name, mode, skip_ratio
        self.height, self.width, self.crop_rows = 480, 640, 40
        self.nr_events_data, self.num_bins = nr_events_data, nr_bins_per_data
        assert nr_events_per_data > 0
        self.nr_events_per_data = nr_events_per_data
        self.event_representation, self.separate_pol, self.normalize_event = event_representation, separate_pol, normalize_event
        self.locations = ['left']
        self.semseg_num_classes, self.augmentation = semseg_num_classes, augmentation
        self.fixed_duration = fixed_duration
        if fixed_duration:
            self.delta_t_us = nr_events_data * delta_t_per_data * 1000
        self.remove_time_window = remove_time_window
        self.require_paired_data = require_paired_data
        self.timestamps = np.loadtxt(str(seq_path / 'semantic' / 'semantic_timestamps.txt'), dtype='int64')[6:]
        if semseg_num_classes not in (11, 19):
            raise ValueError
        label_dir = seq_path / 'semantic' / 'left' / ('11classes' if semseg_num_classes == 11 else '19classes')
        assert label_dir.is_dir()
        self.label_pathstrings = sorted(str(e) for e in label_dir.iterdir() if str(e.name).endswith('.png'))
        assert len(self.label_pathstrings) == self.timestamps.size
        drop = (remove_time_window // 100 + 1) * 2
        self.timestamps = self.timestamps[drop:]
        del self.label_pathstrings[:drop]
        assert len(self.label_pathstrings) == self.timestamps.size
        original_length = len(self.label_pathstrings)
        if skip_ratio != 1:
            new_length = original_length // skip_ratio
            self.timestamps = self.timestamps[:new_length + 1]
            self.label_pathstrings = self.label_pathstrings[:new_length + 1]
            assert len(self.label_pathstrings) == self.timestamps.size
            print("Seq '{}': '{}' of '{}' data loaded with skipping ratio '{}' .".format(
                self.sequence_name, len(self.label_pathstrings), original_length, skip_ratio))
        else:
            print("Seq '{}': '{}' data loaded.".format(self.sequence_name, original_length))
        self.h5f, self.rectify_ev_maps, self.event_slicers = {}, {}, {}
        for location in self.locations:
            d = seq_path / 'events' / location
            self.h5f[location] = open_h5(d / 'events.h5')
            self.event_slicers[location] = EventSlicer(self.h5f[location])
            rect = open_h5(d / 'rectify_map.h5')
            self.rectify_ev_maps[location] = np.ascontiguousarray(np.asarray(rect['rectify_map'][()], dtype=np.float32))
            rect.close()
            assert self.rectify_ev_maps[location].shape == (self.height, self.width, 2)
        self.config_option, self.pl_sources = config_option, pl_sources
        self.superpixel_sources, self.if_sam_distillation = superpixel_sources, if_sam_distillation

    # ------------------------------------------------------------------ reference helpers kept by name
    def getHeightAndWidth(self):
        return self.height, self.width

    @staticmethod
    def get_label(filepath):
        assert Path(filepath).is_file()
        return _io.load_png(str(filepath))

    @staticmethod
    def close_callback(h5f_dict):
        for h5f in h5f_dict.values():
            h5f.close()

    def __len__(self):
        return self.timestamps.size

    def rectify_events(self, x, y, location):
        assert location in self.locations
        rectify_map = self.rectify_ev_maps[location]
        assert x.max() < self.width and y.max() < self.height
        return rectify_map[y, x]

    # ------------------------------------------------------------------ raw event slices (sequence_ov.py:243-307)
    def raw_events(self, index, location='left'):
        """The event rows the reference voxelizes for sample `index`, NOT voxelized: columns + sub-window offsets."""
        ts_end = self.timestamps[index]
        slicer = self.event_slicers[location]
        nwin = self.nr_events_data
        if self.fixed_duration:                                                   # :247-279: nwin windows of equal DURATION
            ts_start = ts_end - self.delta_t_us
            step = self.delta_t_us / nwin
            bounds = []
            for i in range(nwin):
                rng = slicer.window_indices(ts_start + i * step, ts_start + (i + 1) * step)
                if rng is None:
                    raise IndexError(f"{self.sequence_name}[{index}]: time window outside the recording (the reference fails too)")
                bounds.append(rng)
            # consecutive windows [t_i, t_{i+1}) share their boundaries, so the rows are one contiguous range
            a, b = bounds[0][0], bounds[-1][1]
            offs = np.array([bounds[0][0]] + [e for _, e in bounds], dtype=np.int64) - a
            ev = {k: np.array(slicer.events[k][a:b]) for k in ('x', 'y', 'p')}           # one copy out of the (mem-mapped) file
            ev['t'] = np.array(slicer.events['t'][a:b], dtype=np.int64) + slicer.t_offset
        else:                                                                     # :281-305: last N events, nwin equal COUNTS
            nr_events = nwin * self.nr_events_per_data
            rng = slicer.fixed_num_indices(ts_end, nr_events)
            if rng is None:
                raise IndexError(f"{self.sequence_name}[{index}]: label timestamp outside the recording (the reference fails too)")
            a, b = rng                                                            # start_index = 0 / -nr_events rule (:287-290)
            per = (b - a) // nwin                                                 # nr_events_temp = nr_events_loaded // nr_events_data (:302)
            b = a + per * nwin                                                    # the remainder is never voxelized
            offs = np.arange(nwin + 1, dtype=np.int64) * per
            ev = {k: np.array(slicer.events[k][a:b]) for k in ('x', 'y', 'p')}
            ev['t'] = np.array(slicer.events['t'][a:b], dtype=np.int64)           # no t_offset here, as the reference (:95-96)
        return {'x': torch.from_numpy(ev['x'].astype(np.uint16, copy=False)), 'y': torch.from_numpy(ev['y'].astype(np.uint16, copy=False)),
                't': torch.from_numpy(ev['t']), 'p': torch.from_numpy(ev['p'].astype(np.uint8, copy=False)),
                'seg_offsets': torch.from_numpy(offs), 'flip': False}

    # ------------------------------------------------------------------ __getitem__ (sequence_ov.py:225-463)
    def _side_inputs(self, label_path):
        file_path = str(label_path)
        name = Path(label_path).parts[-1]
        frame = recon = None
        if self.config_option in ('frame2voxel', 'frame2recon'):
            p = file_path.replace('/semantic/left/', '/images_aligned/left/')
            frame = _io.image_to_chw_float(p.split('left/')[0] + 'left/' + name)
        if self.config_option in ('recon2voxel', 'frame2recon'):
            p = file_path.replace('/semantic/left/', '/reconstructions/left/')
            recon = _io.image_to_chw_float(p.split('left/')[0] + 'left/' + name)
        return file_path, frame, recon

    # ---- 8-bit maps: host tensors (the reference's path) or, with device_png, the undecoded file bytes
    def _map(self, path):
        if self.device_png:
            return {'png': torch.from_numpy(np.fromfile(str(path), dtype=np.uint8)), 'flip': False,
                    'hw': (self.height - self.crop_rows, self.width)}
        return torch.tensor(_io.load_png(str(path))).squeeze(0).long()

    def _ones_map(self, like):
        if isinstance(like, dict):
            return torch.ones(like['hw'], dtype=torch.int64)
        return torch.ones_like(like)

    @staticmethod
    def _flip_map(m):
        if isinstance(m, dict):
            return dict(m, flip=not m['flip'])
        return torch.flip(m, [1])

    def __getitem__(self, index):
        label_path = self.label_pathstrings[index]
        label_tensor = self._map(label_path) if self.device_png else torch.from_numpy(self.get_label(label_path)).long()
        events = self.raw_events(index) if self.config_option in ('recon2voxel', 'frame2voxel') else None
        file_path, frame, recon = self._side_inputs(label_path)
        if self.mode == 'train':
            pl_path = file_path.replace('semantic/', self.pl_sources + '/').replace('11classes/', '')
            pl = self._map(pl_path)
        else:
            pl = self._ones_map(label_tensor)
        if len(self.superpixel_sources) > 1:
            sp_path = file_path.replace('semantic/', self.superpixel_sources + '/').replace('11classes/', '')
            if self.superpixel_sources.split('_')[1] == 'slic':
                sp_path = sp_path.replace('.png', '_slic_100.png')
            superpixel = self._map(sp_path) if self.device_png else torch.tensor(_io.load_png(sp_path)).long()
        else:
            superpixel = self._ones_map(label_tensor)
        sam_feat = torch.ones((256, 64, 64))
        opt = self.config_option
        if opt not in ('recon2voxel', 'frame2voxel', 'frame2recon', 'recon_only'):
            return None                                                            # the reference falls off the end too
        img = {'recon2voxel': 'recon', 'frame2voxel': 'frame', 'recon_only': 'recon'}.get(opt)
        if self.augmentation:                       # identical order of random.random() / random.uniform / torch.randn draws
            if random.random() >= 0.5:
                if events is not None:
                    events['flip'] = True                                          # torch.flip(event_tensor, [2]) after voxelization
                label_tensor = self._flip_map(label_tensor)
                if recon is not None and opt != 'frame2voxel':
                    recon = torch.flip(recon, [2])
                if frame is not None and opt in ('frame2voxel', 'frame2recon'):
                    frame = torch.flip(frame, [2])
                if opt != 'recon_only':
                    pl = self._flip_map(pl)
                superpixel = self._flip_map(superpixel)
                sam_feat = torch.flip(sam_feat, [2])
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
                x = recon if img == 'recon' else frame
                if random.random() >= 0.5:
                    x = _io.adjust_brightness(x, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    x = _io.adjust_contrast(x, random.uniform(0.8, 1.2))
                if random.random() >= 0.5:
                    x = x + torch.randn(x.size()) * 0.05
                if img == 'recon':
                    recon = x
                else:
                    frame = x
        if opt == 'recon2voxel':
            return events, label_tensor, recon, pl, superpixel, sam_feat, file_path
        if opt == 'frame2voxel':
            return events, label_tensor, frame, pl, superpixel, sam_feat, file_path
        if opt == 'frame2recon':
            return frame, label_tensor, recon, pl, superpixel, sam_feat, file_path
        return label_tensor, recon, superpixel, sam_feat, file_path              # 'recon_only' (:444-463)

    # ------------------------------------------------------------------ GPU voxelization of collated raw events
    def voxelize_batch(self, batch0, device, rectify_maps=None):
        """batch0: collated raw events (datasets/synthetic_events.py:collate) -> B x (nr_events_data*C) x 440 x 640 float32."""
        return voxelize_raw_batch(batch0, device, self.rectify_ev_maps['left'] if rectify_maps is None else rectify_maps,
                                  self.num_bins, self.height, self.width, self.crop_rows, self.nr_events_data)

    def materialize(self, index, device='cuda'):
        """The reference's tuple exactly: voxel tensor (CPU float32) in place of the raw events."""
        from ...datasets.synthetic_events import collate
        item = self[index]
        if isinstance(item[0], dict):
            vox = self.voxelize_batch(collate([item])[0], torch.device(device))[0].cpu()
            return (vox, *item[1:])
        return item


_RMAP_CACHE = {}


def voxelize_raw_batch(batch0, device, rectify_map, C, H, W, crop_rows, nwin):
    from ... import hip
    if torch.is_tensor(rectify_map):
        rmaps = rectify_map
    else:
        key = (id(rectify_map), str(device))
        if key not in _RMAP_CACHE:
            _RMAP_CACHE[key] = torch.from_numpy(np.ascontiguousarray(rectify_map)[None]).to(device)
        rmaps = _RMAP_CACHE[key]
    seg = batch0['seg_offsets']
    B = (seg.numel() - 1) // nwin
    seg_map = batch0.get('seg_map')
    if seg_map is None:
        seg_map = torch.zeros(B * nwin, dtype=torch.int32)
    dev = {k: hip.h2d_async(batch0[k], device) for k in ('x', 'y', 't', 'p')}
    vox = hip.voxelize_dsec_raw(dev['x'], dev['y'], dev['t'], dev['p'], rmaps, hip.h2d_async(seg_map, device), seg, C, H, W,
                                crop_rows=crop_rows)
    vox = vox.view(B, nwin * C, H - crop_rows, W)
    flips = batch0.get('flip')
    if flips is not None and any(flips):
        idx = torch.tensor([i for i, f in enumerate(flips) if f], device=device)
        vox[idx] = torch.flip(vox[idx], [3])
    return vox
