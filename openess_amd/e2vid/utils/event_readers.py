"""Event windows for the offline reconstruction CLI.  Behavioural mirror of the two reader classes of the
reference's e2vid/utils/event_readers.py:8-88; written against their observable semantics, not their text.

File format: a text file whose first line is the sensor size (`width height`) and whose other lines are events `t x y p`
(t in seconds, p in {0, 1}); a `.zip` holding exactly one such member is accepted by the fixed-duration reader (as in the
reference, which asserts on the extension and on a single member).  `start_index` skips that many events after the header.

Both classes are iterators over float64 arrays [n, 4] with columns (t, x, y, p):
  * FixedSizeEventReader: consecutive, non-overlapping windows of `num_events` rows; a shorter last window is delivered;
  * FixedDurationEventReader: a window opens at the stamp of its first event for the first window and at the stamp of the
    previous window's closing event afterwards; it closes with the first event whose stamp exceeds opening stamp + duration, and
    that event is the window's last row.  Events after the last closing event never form a window (the reference runs off the
    end of the file and stops).

Design: the columns are parsed ONCE into one contiguous float64 table with NumPy's C tokenizer (np.loadtxt) and the window
boundaries of the fixed-duration reader are found with binary searches on the time column, so a window costs O(log N) instead
of a Python-level parse of every line; windows are views of the table."""
import io
import os
import zipfile

import numpy as np


def _event_table(path, start_index, allow_zip):
    ext = os.path.splitext(path)[1]
    if allow_zip:
        assert ext in ('.txt', '.zip'), f"event file must be .txt or .zip, got {ext!r}"
    if ext == '.zip':
        with zipfile.ZipFile(path) as zf:
            members = zf.namelist()
            assert len(members) == 1, "the archive must hold exactly one event file"
            text = io.TextIOWrapper(zf.open(members[0], 'r'), encoding='utf-8')
            table = np.loadtxt(text, dtype=np.float64, skiprows=1 + start_index, ndmin=2)
    else:
        table = np.loadtxt(path, dtype=np.float64, skiprows=1 + start_index, ndmin=2)
    if table.size == 0:
        table = np.zeros((0, 4), np.float64)
    if table.shape[1] != 4:
        raise ValueError(f"{path}: expected rows of `t x y p`, found {table.shape[1]} columns")
    return table


class FixedSizeEventReader:
    """Windows of a fixed number of events (variable output frame rate)."""

    def __init__(self, path_to_event_file, num_events=10000, start_index=0):
        print(f"Event windows: {num_events} events each (variable frame rate)")
        self.table = _event_table(path_to_event_file, start_index, allow_zip=False)
        self.num_events = int(num_events)
        self.cursor = 0

    def __iter__(self):
        return self

    def __next__(self):
        lo = self.cursor
        if lo >= len(self.table):
            raise StopIteration
        self.cursor = hi = min(lo + self.num_events, len(self.table))
        return self.table[lo:hi]


class FixedDurationEventReader:
    """Windows of a fixed duration in milliseconds (fixed output frame rate 1000 / duration_ms Hz)."""

    def __init__(self, path_to_event_file, duration_ms=50.0, start_index=0):
        print(f"Event windows: {duration_ms:.2f} ms each ({1000.0 / duration_ms:.1f} Hz)")
        self.table = _event_table(path_to_event_file, start_index, allow_zip=True)
        self.sorted = bool(np.all(np.diff(self.table[:, 0]) >= 0))
        self.duration_s = duration_ms / 1000.0
        self.cursor = 0
        self.last_stamp = None                       # opening stamp of the current window

    def __iter__(self):
        return self

    def __next__(self):
        t = self.table[:, 0]
        lo = self.cursor
        if lo >= len(t):
            raise StopIteration
        if self.last_stamp is None:
            self.last_stamp = float(t[lo])
        limit = self.last_stamp + self.duration_s
        # first row at or after `lo` whose stamp exceeds the limit: a binary search when the stamps are sorted (a recording's
        # are), the linear scan the semantics are defined by otherwise
        if self.sorted:
            close = lo + int(np.searchsorted(t[lo:], limit, side='right'))
        else:
            beyond = np.nonzero(t[lo:] > limit)[0]
            close = lo + int(beyond[0]) if len(beyond) else len(t)
        if close >= len(t):                          # no closing event: the tail is not a window
            self.cursor = len(t)
            raise StopIteration
        self.cursor = close + 1
        self.last_stamp = float(t[close])
        return self.table[lo:close + 1]
