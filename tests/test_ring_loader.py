"""CPU tests of datasets/ring_loader.PinnedRingLoader (the pinned shared ring the loader workers collate into): it must deliver
exactly the batches of DataLoader(..., collate_fn=collate) -- same layout, same values, same order, same epoch length -- while a
slot is reused only after the consumer has let go of it."""
import pytest
import torch

from openess_amd.datasets.ring_loader import Arena, PinnedRingLoader
from openess_amd.datasets.synthetic_events import SyntheticEvents, collate


def _ds(n=12, option='frame2voxel'):
    return SyntheticEvents(length=n, sensor_hw=(60, 80), crop_rows=4, nr_events_data=3, nr_events_window=500, pool=5,
                           config_option=option)


def _same(a, b):
    if torch.is_tensor(a):
        return torch.is_tensor(b) and a.dtype == b.dtype and torch.equal(a, b)
    if isinstance(a, dict):
        return a.keys() == b.keys() and all(_same(a[k], b[k]) for k in a)
    if isinstance(a, (list, tuple)):
        return len(a) == len(b) and all(_same(x, y) for x, y in zip(a, b))
    return a == b


@pytest.mark.parametrize("option", ["frame2voxel", "frame2recon"])
def test_ring_loader_equals_collate_in_order_over_epochs(option):
    ds = _ds(14, option)
    ld = PinnedRingLoader(ds, batch_size=4, shuffle=False, drop_last=False, num_workers=2, slots=3, pin=False)
    try:
        assert len(ld) == 4
        for _ in range(2):                                   # 4 batches through 3 slots, twice: slots are recycled
            got = 0
            for i, b in enumerate(ld):
                idx = list(range(i * 4, min(i * 4 + 4, 14)))
                assert _same(b, collate([ds[j] for j in idx])), i
                got += 1
            assert got == 4
    finally:
        ld.close()


def test_ring_loader_shuffle_drop_last_and_slot_hold():
    ds = _ds(10)
    ld = PinnedRingLoader(ds, batch_size=4, shuffle=True, drop_last=True, num_workers=3, pin=False)
    try:
        assert len(ld) == 2
        torch.manual_seed(7)
        order = torch.randperm(10).tolist()
        torch.manual_seed(7)
        held = []
        for i, b in enumerate(ld):
            ref = collate([ds[j] for j in order[i * 4:i * 4 + 4]])
            assert _same(b, ref)
            held.append((b, ref))
        # batches handed out earlier are still intact while later ones were produced (their slots were not overwritten)
        for b, ref in held:
            assert _same(b, ref)
    finally:
        ld.close()


class _Boom(SyntheticEvents):
    def __getitem__(self, i):
        if i == 5:
            raise ValueError("sample five is broken")
        return super().__getitem__(i)


def test_ring_loader_reports_worker_errors_and_small_slots():
    ds = _Boom(length=8, sensor_hw=(60, 80), crop_rows=4, nr_events_data=3, nr_events_window=500, pool=2)
    ld = PinnedRingLoader(ds, batch_size=4, num_workers=2, pin=False)
    try:
        with pytest.raises(RuntimeError, match="sample five is broken"):
            for _ in ld:
                pass
    finally:
        ld.close()
    ld = PinnedRingLoader(_ds(8), batch_size=4, num_workers=1, slot_bytes=4096, pin=False)
    try:
        with pytest.raises(RuntimeError, match="too small"):
            next(iter(ld))
    finally:
        ld.close()


def test_arena_views_are_aligned_and_disjoint():
    buf = torch.zeros(1 << 16, dtype=torch.uint8)
    a = Arena(buf)
    x = a.cat([torch.arange(5, dtype=torch.int64), torch.arange(3, dtype=torch.int64)])
    y = a.stack([torch.ones(2, 3), torch.zeros(2, 3)])
    z = a.put(torch.tensor([1, 2, 3], dtype=torch.uint8))
    assert x.start % 256 == 0 and y.start % 256 == 0 and z.start % 256 == 0 and x.start < y.start < z.start
    assert x.t.tolist() == [0, 1, 2, 3, 4, 0, 1, 2] and y.t.shape == (2, 2, 3) and z.t.tolist() == [1, 2, 3]
    assert x.t.data_ptr() == buf.data_ptr() + x.start


def test_ring_loader_abandoned_iteration_does_not_leak_into_the_next():
    ds = _ds(16)
    ld = PinnedRingLoader(ds, batch_size=4, shuffle=False, num_workers=2, slots=3, pin=False)
    try:
        it = iter(ld)
        first = next(it)                      # the workers are already filling the other slots for batches 1, 2
        assert _same(first, collate([ds[j] for j in range(4)]))
        del it                                # consumer walks away mid-epoch
        for rep in range(2):
            n = 0
            for i, b in enumerate(ld):        # (a batch is only valid until the next one is requested: 3 slots, 4 batches)
                assert _same(b, collate([ds[j] for j in range(i * 4, i * 4 + 4)])), (rep, i)
                n += 1
            assert n == 4
    finally:
        ld.close()
