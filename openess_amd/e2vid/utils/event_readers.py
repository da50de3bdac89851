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

Design: the file is parsed in BOUNDED chunks (CHUNK_ROWS lines at a time through NumPy's C tokenizer, `np.loadtxt(handle,
max_rows=...)` on one open handle) into a float64 table that only ever holds the unconsumed rows plus one chunk, so a multi-GB
recording costs CHUNK_ROWS x 32 B of memory and the first window is delivered after the first chunk, as the reference's
streaming readers do (pandas `chunksize` / line iteration).  The window boundaries of the fixed-duration reader are found with
binary searches on the time column of the buffered rows (a linear scan when the stamps are not sorted); the search state (opening
stamp, rows already searched) carries across chunk refills."""
import io
import os
import warnings
import zipfile

import numpy as np

CHUNK_ROWS = 1 << 20            # 32 MB of float64 rows per refill


class _EventStream:
    """Rows `t x y p` of an event text file as a sliding float64 buffer: `rows` holds the unconsumed rows read so far."""

    def __init__(self, path, start_index, allow_zip, chunk_rows=None):
        ext = os.path.splitext(path)[1]
        if allow_zip:
            assert ext in ('.txt', '.zip'), f"event file must be .txt or .zip, got {ext!r}"
        self.path, self._zip = path, None
        if ext == '.zip':
            self._zip = zipfile.ZipFile(path)
            members = self._zip.namelist()
            assert len(members) == 1, "the archive must hold exactly one event file"
            self.handle = io.TextIOWrapper(self._zip.open(members[0], 'r'), encoding='utf-8')
        else:
            self.handle = open(path, 'r')
        for _ in range(1 + start_index):             # sensor-size header + skipped events
            if not self.handle.readline():
                break
        self.chunk_rows = int(chunk_rows or CHUNK_ROWS)
        self.rows = np.zeros((0, 4), np.float64)
        self.eof = False
        self.sorted = True                           # every stamp read so far is >= its predecessor (checked once per chunk)
        self._last_t = -np.inf

    def refill(self):
        """Append up to chunk_rows more rows; False at the end of the file."""
        if self.eof:
            return False
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")          # "input contained no data" at the end of the file
            chunk = np.loadtxt(self.handle, dtype=np.float64, max_rows=self.chunk_rows, ndmin=2)
        if chunk.size == 0:
            self.eof = True
            self.close()
            return False
        if chunk.shape[1] != 4:
            raise ValueError(f"{self.path}: expected rows of `t x y p`, found {chunk.shape[1]} columns")
        if len(chunk) < self.chunk_rows:
            self.eof = True
            self.close()
        if self.sorted:
            tc = chunk[:, 0]
            self.sorted = bool(tc[0] >= self._last_t and np.all(tc[1:] >= tc[:-1]))
        self._last_t = float(chunk[-1, 0])
        self.rows = chunk if len(self.rows) == 0 else np.concatenate([self.rows, chunk])
        return True

    def consume(self, n):
        """The first n buffered rows (a copy-free view of the current buffer), dropped from the buffer."""
        out, self.rows = self.rows[:n], self.rows[n:]
        return out

    def close(self):
        if self.handle is not None:
            self.handle.close()
            self.handle = None
        if self._zip is not None:
            self._zip.close()
            self._zip = None


class FixedSizeEventReader:
    """Windows of a fixed number of events (variable output frame rate)."""

    def __init__(self, path_to_event_file, num_events=10000, start_index=0, chunk_rows=None):
        print(f"Event windows: {num_events} events each (variable frame rate)")
        self.stream = _EventStream(path_to_event_file, start_index, allow_zip=False, chunk_rows=chunk_rows)
        self.num_events = int(num_events)

    def __iter__(self):
        return self

    def __next__(self):
        st = self.stream
        while len(st.rows) < self.num_events and st.refill():
            pass
        if len(st.rows) == 0:
            raise StopIteration
        return st.consume(min(self.num_events, len(st.rows)))


class FixedDurationEventReader:
    """Windows of a fixed duration in milliseconds (fixed output frame rate 1000 / duration_ms Hz)."""

    def __init__(self, path_to_event_file, duration_ms=50.0, start_index=0, chunk_rows=None):
        print(f"Event windows: {duration_ms:.2f} ms each ({1000.0 / duration_ms:.1f} Hz)")
        self.stream = _EventStream(path_to_event_file, start_index, allow_zip=True, chunk_rows=chunk_rows)
        self.duration_s = duration_ms / 1000.0
        self.last_stamp = None                       # opening stamp of the current window

    def __iter__(self):
        return self

    def __next__(self):
        st = self.stream
        if len(st.rows) == 0 and not st.refill():
            raise StopIteration
        if self.last_stamp is None:
            self.last_stamp = float(st.rows[0, 0])
        limit = self.last_stamp + self.duration_s
        searched = 0                                 # buffered rows already known not to close the window
        while True:
            t = st.rows[searched:, 0]
            # first row whose stamp exceeds the limit: a binary search when this stretch is sorted (a recording's stamps are),
            # the linear scan the semantics are defined by otherwise
            if st.sorted:
                k = int(np.searchsorted(t, limit, side='right'))
            else:
                beyond = np.nonzero(t > limit)[0]
                k = int(beyond[0]) if len(beyond) else len(t)
            if k < len(t):
                close = searched + k
                break
            searched = len(st.rows)
            if not st.refill():                      # no closing event: the tail is not a window
                st.consume(len(st.rows))
                raise StopIteration
        self.last_stamp = float(st.rows[close, 0])
        return st.consume(close + 1)
