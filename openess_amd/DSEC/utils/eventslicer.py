"""EventSlicer (DSEC/utils/eventslicer.py:10-203) over DSEC's `events.h5` -- same methods, same index arithmetic.

Two differences, both on purpose:
  * the reference's numba scan `get_time_indices_offsets` (:152-203, linear from both ends of the conservative
    millisecond window) is a BINARY SEARCH here (np.searchsorted 'left' is exactly its post-condition
    `t[idx-1] < t_us <= t[idx]`);
  * containers: h5py when it is importable, otherwise the unpacked layout `<name>_h5/<dataset>.npy` (memory-mapped; made
    by tools/dsec_unpack_h5.py) -- slices of a memmap copy straight into the pinned batch buffers, no HDF5 chunk cache.
"""
import math
import os

import numpy as np


class _NpyGroup(dict):
    """Read-only stand-in for an h5py.File over `<path minus .h5>_h5/*.npy` (keys use '/' like HDF5)."""

    def __init__(self, directory):
        super().__init__()
        for f in sorted(os.listdir(directory)):
            if f.endswith(".npy"):
                self[f[:-4]] = np.load(os.path.join(directory, f), mmap_mode='r')

    def __getitem__(self, k):
        return dict.__getitem__(self, k.replace("/", "_"))

    def __contains__(self, k):
        return dict.__contains__(self, k.replace("/", "_"))

    def close(self):
        pass


def open_h5(path):
    """events.h5 / rectify_map.h5 -> mapping of datasets.  Prefers the unpacked .npy layout; falls back to h5py."""
    path = str(path)
    unpacked = path[:-3] + "_h5" if path.endswith(".h5") else path
    if os.path.isdir(unpacked):
        return _NpyGroup(unpacked)
    try:
        import h5py
        try:
            import hdf5plugin  # noqa: F401  (registers the blosc filter DSEC uses)
        except ImportError:
            pass
    except ImportError as e:
        raise ImportError(f"{path}: reading DSEC HDF5 needs h5py (+ hdf5plugin), which is not installed, and the unpacked layout "
                          f"{unpacked}/ does not exist.  Run tools/dsec_unpack_h5.py on a machine with h5py.") from e
    return h5py.File(path, 'r')


class EventSlicer:
    def __init__(self, h5f):
        self.h5f = h5f
        self.events = {k: h5f['events/{}'.format(k)] for k in ('p', 'x', 'y', 't')}
        self.ms_to_idx = np.asarray(h5f['ms_to_idx'], dtype='int64')
        self.t_offset = int(np.asarray(h5f['t_offset'])[()]) if 't_offset' in h5f else 0
        self.t_final = int(self.events['t'][-1]) + self.t_offset

    def get_start_time_us(self):
        return self.t_offset

    def get_final_time_us(self):
        return self.t_final

    @staticmethod
    def get_conservative_window_ms(ts_start_us, ts_end_us):
        assert ts_end_us > ts_start_us
        return math.floor(ts_start_us / 1000), math.ceil(ts_end_us / 1000)

    @staticmethod
    def get_conservative_ms(ts_us):
        return math.floor(ts_us / 1000), math.ceil(ts_us / 1000)

    @staticmethod
    def get_time_indices_offsets(time_array, time_start_us, time_end_us):
        """(idx_start, idx_end) with time_start_us <= time_array[idx_start:idx_end] < time_end_us (eventslicer.py:152-203)."""
        assert time_array.ndim == 1
        n = time_array.size
        if n == 0 or time_array[-1] < time_start_us:      # (the reference raises IndexError on an empty window)
            return n, n
        return int(np.searchsorted(time_array, time_start_us, side='left')), int(np.searchsorted(time_array, time_end_us, side='left'))

    def ms2idx(self, time_ms):
        assert time_ms >= 0
        if time_ms >= self.ms_to_idx.size:
            return None
        return self.ms_to_idx[time_ms]

    def window_indices(self, t_start_us, t_end_us):
        """Absolute event index range of [t_start_us, t_end_us) -- the slicing half of get_events (:32-66)."""
        assert t_start_us < t_end_us
        t_start_us -= self.t_offset
        t_end_us -= self.t_offset
        ms0, ms1 = self.get_conservative_window_ms(t_start_us, t_end_us)
        i0, i1 = self.ms2idx(ms0), self.ms2idx(ms1)
        if i0 is None or i1 is None:
            return None
        window = np.asarray(self.events['t'][i0:i1])
        a, b = self.get_time_indices_offsets(window, t_start_us, t_end_us)
        return int(i0 + a), int(i0 + b)

    def get_events(self, t_start_us, t_end_us, max_events_per_data=-1):
        rng = self.window_indices(t_start_us, t_end_us)
        if rng is None:
            print('Error', 'start', t_start_us, 'end', t_end_us)
            return None
        a, b = rng
        events = {'t': np.asarray(self.events['t'][a:b]) + self.t_offset}
        for k in ('p', 'x', 'y'):
            events[k] = np.asarray(self.events[k][a:b])
        return events

    def fixed_num_indices(self, t_end_us, nr_events=100000):
        """Absolute index range of the last `nr_events` events before t_end_us (:68-98)."""
        t_end_us -= self.t_offset
        lo_ms, hi_ms = self.get_conservative_ms(t_end_us)
        i0, i1 = self.ms2idx(lo_ms), self.ms2idx(hi_ms)
        if i0 is None or i1 is None:
            return None
        window = np.asarray(self.events['t'][i0:i1])
        _, off = self.get_time_indices_offsets(window, t_end_us, t_end_us)
        end = int(i0 + off)
        return max(end - nr_events, 0), end

    def get_events_fixed_num(self, t_end_us, nr_events=100000):
        rng = self.fixed_num_indices(t_end_us, nr_events)
        if rng is None:
            return None
        a, b = rng
        return {k: np.asarray(self.events[k][a:b]) for k in self.events}          # NOTE: 't' without t_offset, as the reference

    def get_events_fixed_num_recurrent(self, t_start_us_idx, t_end_us_idx):
        assert t_start_us_idx < t_end_us_idx
        return {k: np.asarray(self.events[k][t_start_us_idx:t_end_us_idx]) for k in self.events}
