#!/usr/bin/env python3
"""Golden vectors for the offline-reconstruction event readers, produced by RUNNING the reference's own
`FixedSizeEventReader` / `FixedDurationEventReader` (e2vid/utils/event_readers.py:8-88, imported from /root/reference) on a
seeded synthetic event text file.  Stored: the event table that was written to the file and, per reader configuration, the
row range [first, last] of every window the reference delivered (the windows are contiguous row ranges of the table, which the
script asserts), so the fixture is numeric data only.
Run:  python tests/golden/gen_golden_event_readers.py"""
import os
import sys
import tempfile
import zipfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def synth_table(rng, n):
    # bursty stamps with ties and long gaps so that fixed-duration windows of very different lengths occur
    dt = rng.exponential(0.004, n) * (rng.random(n) > 0.15)
    dt[rng.integers(0, n, 6)] += 0.2
    t = np.round(0.5 + np.cumsum(dt), 6)
    return np.stack([t, rng.integers(0, 64, n), rng.integers(0, 48, n), rng.integers(0, 2, n)], 1).astype(np.float64)


def write_txt(path, table):
    with open(path, "w") as f:
        f.write("64 48\n")
        for t, x, y, p in table:
            f.write(f"{t:.6f} {int(x)} {int(y)} {int(p)}\n")


def locate(table, start, win):
    """Row range of `win` inside `table` (windows are consecutive)."""
    n = len(win)
    assert np.array_equal(np.asarray(win, np.float64), table[start:start + n]), "window is not the next contiguous row range"
    return start, start + n - 1


def main():
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from e2vid.utils.event_readers import FixedDurationEventReader, FixedSizeEventReader
    rng = np.random.default_rng(1205)
    table = synth_table(rng, 700)
    out = {"table": table}
    with tempfile.TemporaryDirectory() as d:
        txt = os.path.join(d, "events.txt")
        write_txt(txt, table)
        zp = os.path.join(d, "events.zip")
        with zipfile.ZipFile(zp, "w") as z:
            z.write(txt, "events.txt")
        for n, start in ((100, 0), (64, 37), (700, 0), (1000, 5)):
            ranges, pos = [], start
            for w in FixedSizeEventReader(txt, num_events=n, start_index=start):
                a, b = locate(table, pos, w)
                ranges.append((a, b))
                pos = b + 1
            out[f"size_{n}_{start}"] = np.array(ranges, np.int64).reshape(-1, 2)
        for ms, start, path in ((50.0, 0, txt), (10.0, 3, txt), (250.0, 0, zp), (5000.0, 0, txt)):
            ranges, pos = [], start
            for w in FixedDurationEventReader(path, duration_ms=ms, start_index=start):
                a, b = locate(table, pos, w)
                ranges.append((a, b))
                pos = b + 1
            out[f"dur_{int(ms)}_{start}_{os.path.splitext(path)[1][1:]}"] = np.array(ranges, np.int64).reshape(-1, 2)
    np.savez_compressed(os.path.join(HERE, "event_readers.npz"), **out)
    for k, v in out.items():
        print(k, v.shape)


if __name__ == "__main__":
    main()
