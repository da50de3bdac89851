"""Synthetic provider with the reference's dataset tuple contract.

`__getitem__` returns the DSEC-style 7-tuple
    (event_or_frame, label, frame_or_recon, pl, superpixel, sam_feat, file_path)
(DSEC/dataset/sequence_ov.py:384,409,440).  In the voxel options the first item is NOT a pre-built voxel
tensor but a dict of the sample's RAW event columns (x,y: uint16, t: int64 us, p: uint8): the voxelizer runs
on the GPU for the whole batch at once (trainer `prepare_batch`), which is the point of the MI355X path.
`collate` stacks everything and builds the segment offsets."""
import numpy as np
import torch
from torch.utils.data import Dataset

from . import _synth


class SyntheticEvents(Dataset):
    def __init__(self, length=16, sensor_hw=(480, 640), crop_rows=40, nr_events_data=20, nr_events_window=100000,
                 nr_bins=5, num_classes=11, config_option='frame2voxel', superpixel_size=100, mode='train', seed=1205, pool=0):
        self.length, self.sensor_hw, self.crop_rows = length, tuple(sensor_hw), crop_rows
        self.nr_events_data, self.nr_events_window, self.nr_bins = nr_events_data, nr_events_window, nr_bins
        self.num_classes, self.config_option, self.superpixel_size = num_classes, config_option, superpixel_size
        self.mode, self.seed = mode, seed
        self.require_paired_data = False
        self.rectify_map = _synth.rectify_map(*self.sensor_hw)
        # pool > 0: `pool` distinct samples are generated ONCE here (before the DataLoader forks its workers: the arrays are
        # shared copy-on-write) and index i serves sample i % pool.  A worker then costs what a real loader costs -- copying
        # a sample's raw columns out of a memory map -- instead of drawing 2 M random events per sample (~0.3 s of NumPy),
        # so tools/bench_train_loop.py measures the ingest pipeline and not the random-number generator.
        self.pool = int(pool)
        self._pool = [self._make(i) for i in range(self.pool)] if self.pool > 0 else None

    def __len__(self):
        return self.length

    def getHeightAndWidth(self):
        return self.sensor_hw[0] - self.crop_rows, self.sensor_hw[1]

    def __getitem__(self, index):
        if self._pool is not None:
            item = self._pool[index % self.pool]
            return (*item[:-1], f"synthetic/{self.mode}/{index:06d}")
        return self._make(index)

    def _make(self, index):
        H, W = self.sensor_hw
        Hn = H - self.crop_rows
        rng = np.random.default_rng(self.seed + index + (0 if self.mode == 'train' else 10**6))
        n = self.nr_events_data * self.nr_events_window
        x, y, t, p = _synth.dsec_raw_events(n, H, W, seed=self.seed + index)
        events = {'x': torch.from_numpy(x), 'y': torch.from_numpy(y), 't': torch.from_numpy(t), 'p': torch.from_numpy(p)}
        frame = torch.from_numpy(rng.uniform(0, 1, (3, Hn, W)).astype(np.float32))
        label = rng.integers(0, self.num_classes, (Hn, W))
        label[rng.uniform(0, 1, (Hn, W)) < 0.05] = 255
        label = torch.from_numpy(label).long()
        pl = label.clone()
        g = int(round(self.superpixel_size ** 0.5))
        yy = (np.arange(Hn) * g // Hn)[:, None]
        xx = (np.arange(W) * g // W)[None, :]
        superpixel = torch.from_numpy((yy * g + xx).astype(np.int64))
        sam_feat = torch.ones(256, 64, 64)
        first = events if self.config_option in ('frame2voxel', 'recon2voxel') else frame
        return first, label, frame, pl, superpixel, sam_feat, f"synthetic/{self.mode}/{index:06d}"


def collate(samples, arena=None):
    """Batch the reference-shaped tuples (6 items for DDD17, 7 for DSEC; the last one is the file path) into ONE layout,
    the 7-slot DSEC one with sam_feat = None for DDD17.  Raw-event dicts
    in slot 0 are concatenated into SoA columns with per-sub-window segment offsets (pinned by DataLoader(pin_memory=True)).
    arena: datasets.ring_loader.Arena -- the large tensors are then written straight into a slot of the pinned loader ring
    (`cat` / `stack` with `out=` views of the slot) instead of into fresh allocations."""
    cat = torch.cat if arena is None else arena.cat
    stk = torch.stack if arena is None else arena.stack
    first = [s[0] for s in samples]
    if isinstance(first[0], dict) and 'events' in first[0]:                       # DDD17: int64 [N,4] rows per sample
        batch0 = {'events_list': [f['events'] if arena is None else arena.put(f['events']) for f in first], 'flip': [bool(f.get('flip', False)) for f in first]}
    elif isinstance(first[0], dict):                                              # DSEC / synthetic: x, y, t, p columns
        ev = {k: cat([f[k] for f in first]) for k in ('x', 'y', 't', 'p')}
        ev['events_per_sample'] = torch.tensor([f['x'].numel() for f in first])
        if 'seg_offsets' in first[0]:
            offs, base = [torch.zeros(1, dtype=torch.int64)], 0
            for f in first:
                offs.append(f['seg_offsets'][1:].to(torch.int64) + base)
                base += int(f['x'].numel())
            ev['seg_offsets'] = torch.cat(offs)
            ev['flip'] = [bool(f.get('flip', False)) for f in first]
            if 'sequence' in first[0]:
                ev['sequence'] = [int(f['sequence']) for f in first]
        batch0 = ev
    else:
        batch0 = stk(first)
    n = len(samples[0])
    if n not in (6, 7):
        raise ValueError(f"dataset tuples have 6 (DDD17) or 7 (DSEC) items, got {n}")
    def stack(i):
        col = [s[i] for s in samples]
        if isinstance(col[0], dict) and 'png' in col[0]:      # undecoded 8-bit maps (device_png): the files back to back
            return {'png_bytes': cat([c['png'] for c in col]), 'png_lengths': [int(c['png'].numel()) for c in col],
                    'flip': [bool(c['flip']) for c in col], 'hw': tuple(col[0]['hw'])}
        return stk(col)
    rest = [stack(i) for i in range(1, n - 1)]
    if n == 6:                     # DDD17 has no sam_feat (ddd17_events_loader.py:290): the batch ALWAYS carries the 7-slot layout
        rest = rest[:4] + [None]   # (first, label, frame | recon, pl, superpixel, sam_feat | None, file_paths); nothing downstream guesses
    return (batch0, *rest, [s[n - 1] for s in samples])
