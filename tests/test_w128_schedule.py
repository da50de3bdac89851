"""CPU test of the static tile lists of the persistent ConvLSTM kernel (csrc/conv_lstm_w128.h, host part in conv_fwd.hip): every tile
of every problem exactly once, lists balanced to within one long tile, the XCD chunking of the one-tile kernels."""
import ctypes

import numpy as np
import pytest

from openess_amd import _lib


def _lists(tm, tn, cin, grid=256):
    lib = _lib.load()
    n = len(tm)
    arr = lambda v: (ctypes.c_int * n)(*v)      # noqa: E731
    stride = ctypes.c_int(0)
    a, b, c = arr(tm), arr(tn), arr(cin)
    rc = lib.oess_convlstm_w128_tile_lists(a, b, c, n, grid, None, 0, ctypes.byref(stride))
    if rc != 0:
        return rc, None
    buf = (ctypes.c_int * (grid * stride.value))()
    assert lib.oess_convlstm_w128_tile_lists(a, b, c, n, grid, buf, grid * stride.value, ctypes.byref(stride)) == 0
    return 0, np.frombuffer(buf, dtype=np.int32).reshape(grid, stride.value).copy()


@pytest.mark.parametrize("tm,tn,cin", [
    ([138, 550, 2200], [4, 2, 1], [512, 256, 128]),        # the BASELINE launch: levels 2, 1, 0 (longest K first)
    ([550, 2200], [2, 1], [256, 128]),                     # a drain launch of the skewed schedule
    ([2200], [1], [64]),                                   # first sub-window of level 0 (x half only)
    ([3, 5, 1], [1, 2, 4], [128, 128, 256]),               # fewer tiles than workgroups
])
def test_every_tile_exactly_once_and_balanced(tm, tn, cin):
    rc, L = _lists(tm, tn, cin)
    assert rc == 0
    seen = {}
    cost = np.zeros(L.shape[0])
    for b in range(L.shape[0]):
        row = L[b]
        end = np.where(row < 0)[0]
        assert len(end) > 0, "a list must be terminated"
        k = int(end[0])
        assert (row[k:] == -1).all()
        for e in row[:k]:
            p, t = int(e) >> 24, int(e) & 0xffffff
            assert 0 <= p < len(tm) and 0 <= t < tm[p] * tn[p]
            seen[(p, t)] = seen.get((p, t), 0) + 1
            cost[b] += cin[p] // 64 * 9 * 2330 + 20000
            # XCD chunking: tile t of problem p belongs to the XCD of its contiguous eighth (as conv3x3_halo_group_kernel)
            nwg = tm[p] * tn[p]
            q, r = nwg >> 3, nwg & 7
            x = b & 7
            base = x * (q + 1) if x < r else r * (q + 1) + (x - r) * q
            assert base <= t < base + q + (1 if x < r else 0)
    assert len(seen) == sum(m * n for m, n in zip(tm, tn)) and set(seen.values()) == {1}
    longest_tile = max(c // 64 * 9 * 2330 + 20000 for c in cin)
    for x in range(8):
        cx = cost[x::8]
        assert cx.max() - cx.min() <= longest_tile          # greedy onto the least loaded workgroup


def test_bad_arguments_and_capacity():
    assert _lists([2200], [1], [128], grid=250)[0] != 0     # grid must be a multiple of 8
    rc, _ = _lists([200000], [4], [128])                    # more than 124 tiles per workgroup
    assert rc != 0
