#!/usr/bin/env python3
"""Unpack DSEC's `events/<loc>/events.h5` and `rectify_map.h5` into the memory-mappable layout
`<name>_h5/<dataset>.npy` that openess_amd.DSEC.utils.eventslicer reads without h5py:

    python tools/dsec_unpack_h5.py /data/DSEC            # walks train/*/events/left and test/*/events/left

Needs h5py + hdf5plugin (DSEC compresses with blosc) on the machine that runs it; the training machine then needs neither.
One-off cost: the unpacked event columns are 13 bytes/event (x,y: uint16, p: uint8, t: int64)."""
import os
import sys

import numpy as np


def unpack(h5_path, names):
    import h5py
    try:
        import hdf5plugin  # noqa: F401
    except ImportError:
        pass
    out = h5_path[:-3] + "_h5"
    os.makedirs(out, exist_ok=True)
    with h5py.File(h5_path, 'r') as f:
        for name in names:
            if name not in f:
                continue
            arr = np.asarray(f[name][()])
            if name == 'events/t':
                arr = arr.astype(np.int64)
            np.save(os.path.join(out, name.replace('/', '_') + '.npy'), arr)
            print(h5_path, name, arr.shape, arr.dtype)


def main(root):
    for split in ('train', 'test'):
        d = os.path.join(root, split)
        for seq in sorted(os.listdir(d)) if os.path.isdir(d) else []:
            for loc in ('left',):
                ev = os.path.join(d, seq, 'events', loc)
                if os.path.isfile(os.path.join(ev, 'events.h5')):
                    unpack(os.path.join(ev, 'events.h5'), ['events/p', 'events/x', 'events/y', 'events/t', 'ms_to_idx', 't_offset'])
                if os.path.isfile(os.path.join(ev, 'rectify_map.h5')):
                    unpack(os.path.join(ev, 'rectify_map.h5'), ['rectify_map'])


if __name__ == "__main__":
    main(sys.argv[1])
