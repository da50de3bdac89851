"""Event-window readers of e2vid/utils/event_readers.py:8-88 for the text format `run_reconstruction.py` consumes
(first line `width height`, then one `t x y p` row per event; `.zip` with a single member also accepted by the
fixed-duration reader).  Same window semantics: FixedSizeEventReader = non-overlapping windows of N rows (the last one
may be shorter); FixedDurationEventReader closes a window with the first event whose stamp exceeds window start +
duration (that event is the LAST row of the window, as in the reference)."""
import zipfile
from os.path import splitext

import numpy as np


class FixedSizeEventReader:
    def __init__(self, path_to_event_file, num_events=10000, start_index=0):
        import pandas as pd
        print('Will use fixed size event windows with {} events'.format(num_events))
        print('Output frame rate: variable')
        self.iterator = pd.read_csv(path_to_event_file, sep=r'\s+', header=None, names=['t', 'x', 'y', 'pol'],
                                    dtype={'t': np.float64, 'x': np.int16, 'y': np.int16, 'pol': np.int16}, engine='c',
                                    skiprows=start_index + 1, chunksize=num_events, nrows=None, memory_map=True)

    def __iter__(self):
        return self

    def __next__(self):
        return self.iterator.__next__().values          # float64 [n, 4] (t, x, y, p), like DataFrame.values in the reference


class FixedDurationEventReader:
    def __init__(self, path_to_event_file, duration_ms=50.0, start_index=0):
        print('Will use fixed duration event windows of size {:.2f} ms'.format(duration_ms))
        print('Output frame rate: {:.1f} Hz'.format(1000.0 / duration_ms))
        ext = splitext(path_to_event_file)[1]
        assert ext in ['.txt', '.zip']
        self.is_zip_file = ext == '.zip'
        if self.is_zip_file:
            self.zip_file = zipfile.ZipFile(path_to_event_file)
            names = self.zip_file.namelist()
            assert len(names) == 1
            self.event_file = self.zip_file.open(names[0], 'r')
        else:
            self.event_file = open(path_to_event_file, 'r')
        for _ in range(1 + start_index):
            self.event_file.readline()
        self.last_stamp = None
        self.duration_s = duration_ms / 1000.0

    def __iter__(self):
        return self

    def __del__(self):
        if getattr(self, 'is_zip_file', False):
            self.zip_file.close()
        if hasattr(self, 'event_file'):
            self.event_file.close()

    def __next__(self):
        event_list = []
        for line in self.event_file:
            if self.is_zip_file:
                line = line.decode("utf-8")
            t, x, y, pol = line.split(' ')
            t, x, y, pol = float(t), int(x), int(y), int(pol)
            event_list.append([t, x, y, pol])
            if self.last_stamp is None:
                self.last_stamp = t
            if t > self.last_stamp + self.duration_s:
                self.last_stamp = t
                return np.array(event_list)
        raise StopIteration
