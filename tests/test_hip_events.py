"""GPU parity: HIP voxelizers / normaliser (through the C-ABI) vs the oracle and the committed
golden vectors.  Indices are checked bit-exactly (count mode, integer-coordinate fixtures);
float values within an absolute tolerance that covers only the LDS-atomic summation order."""
import numpy as np
import pytest
import torch

from oracle import events as oe
from tests import synth

pytestmark = pytest.mark.gpu
ATOL = 2e-5      # |sum| stays O(10) at these densities; fp32 order-of-summation noise only


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def seg(*lens):
    return torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int64)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_trilinear_golden(golden_events, tag):
    from openess_amd import hip
    g = golden_events
    C, H, W = (int(v) for v in g[f"tri_{tag}_chw"])
    x, y, p, t = (dev(g[f"tri_{tag}_{k}"]) for k in "xypt")
    out = hip.voxelize_trilinear(x, y, p, t, seg(x.numel()), C, H, W).cpu().numpy()
    np.testing.assert_allclose(out, g[f"tri_{tag}_out_norm0"], rtol=0, atol=ATOL)
    # indices: count mode vs oracle count mode, exact
    cnt = hip.voxelize_trilinear(x, y, p, t, seg(x.numel()), C, H, W, count_mode=True).cpu().numpy()
    ref = oe.voxelgrid_trilinear(g[f"tri_{tag}_x"], g[f"tri_{tag}_y"], g[f"tri_{tag}_p"], g[f"tri_{tag}_t"], C, H, W,
                                 count_mode=True)
    assert np.array_equal(cnt, ref)


def test_trilinear_integer_coords_bit_exact(golden_events):
    from openess_amd import hip
    g = golden_events
    C, H, W = (int(v) for v in g["tri_int_chw"])
    x, y, p, t = (dev(g[f"tri_int_{k}"]) for k in "xypt")
    out = hip.voxelize_trilinear(x, y, p, t, seg(x.numel()), C, H, W).cpu().numpy()
    assert np.array_equal(out, g["tri_int_out"])


def test_trilinear_segments_crop_and_edges():
    """Ragged segments incl. empty, single-event (0/0 -> NaN time -> nothing) and all-equal timestamps."""
    from openess_amd import hip
    rng = np.random.default_rng(3)
    C, H, W, crop = 5, 70, 150, 6
    lens = [4000, 0, 1, 2500, 37, 3000]
    N = sum(lens)
    x = rng.uniform(-2, W + 1, N).astype(np.float32)
    y = rng.uniform(-2, H + 1, N).astype(np.float32)
    p = rng.integers(0, 2, N).astype(np.float32)
    t = np.empty(N, np.float32)
    off = np.concatenate([[0], np.cumsum(lens)])
    for i, n in enumerate(lens):
        if n:
            tt = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
            tt[0], tt[-1] = 0, 1 if n > 1 else 0
            t[off[i]:off[i + 1]] = tt
    t[off[4]:off[5]] = 0.25          # all-equal timestamps in segment 4
    for cm in (False, True):
        out = hip.voxelize_trilinear(dev(x), dev(y), dev(p), dev(t), seg(*lens), C, H, W, crop_rows=crop,
                                     count_mode=cm).cpu().numpy()
        assert out.shape == (len(lens) * C, H - crop, W)
        for i, n in enumerate(lens):
            s, e = off[i], off[i + 1]
            ref = np.zeros((C, H, W), np.float32) if n == 0 else \
                oe.voxelgrid_trilinear(x[s:e], y[s:e], p[s:e], t[s:e], C, H, W, count_mode=cm)
            ref = ref[:, :H - crop]
            got = out[i * C:(i + 1) * C]
            if cm:
                assert np.array_equal(got, ref), i
            else:
                np.testing.assert_allclose(got, ref, rtol=0, atol=ATOL)
    assert not out[C:2 * C].any() and not out[2 * C:3 * C].any() and not out[4 * C:5 * C].any()


def test_trilinear_dense_tiles_long_segments_and_two_bins():
    """The splat's less-travelled paths against the oracle: (a) tiles far denser than the 32-bit accumulator bound (64-bit
    fixed point in two half-height passes, more than 768 records per tile -> the per-lane run search), with large |value|;
    (b) a segment longer than 64 sort slices (> 131 072 events: run table walked in 64-slice chunks); (c) C = 2 (32-row
    tiles: the strided write-out) and C = 11.  The float64 sums are the yardstick for the dense case: the oracle's own
    sequential fp32 accumulation carries an error that grows with the count, the fixed-point sum does not."""
    from openess_amd import hip
    rng = np.random.default_rng(11)

    def one(C, H, W, lens, xr, yr, pvals, atol, rtol):
        N = sum(lens)
        x = rng.uniform(xr[0], xr[1], N).astype(np.float32)
        y = rng.uniform(yr[0], yr[1], N).astype(np.float32)
        p = rng.choice(np.asarray(pvals, np.float32), N)
        t = np.empty(N, np.float32)
        off = np.concatenate([[0], np.cumsum(lens)])
        for i, n in enumerate(lens):
            tt = np.sort(rng.uniform(0, 1, n)).astype(np.float32)
            tt[0], tt[-1] = 0, 1
            t[off[i]:off[i + 1]] = tt
        for cm in (True, False):
            out = hip.voxelize_trilinear(dev(x), dev(y), dev(p), dev(t), seg(*lens), C, H, W, count_mode=cm).cpu().numpy()
            for i, n in enumerate(lens):
                s, e = off[i], off[i + 1]
                ref = oe.voxelgrid_trilinear(x[s:e], y[s:e], p[s:e], t[s:e], C, H, W, count_mode=cm)
                got = out[i * C:(i + 1) * C]
                if cm:
                    assert np.array_equal(got, ref), (C, H, W, i)
                else:
                    np.testing.assert_allclose(got, ref, rtol=rtol, atol=atol)

    # (a) 6 000 and 900 events inside a 40 x 12 pixel patch (one or two tiles), values 2p-1 in {-1, 1, 199, -201}
    one(5, 64, 128, [6000, 900], (30.0, 70.0), (10.0, 22.0), [0.0, 1.0, 100.0, -100.0], atol=2e-2, rtol=2e-5)
    # (b) 150 000 events in one segment (74 slices) + a short one, small image so that the oracle stays quick
    one(5, 48, 128, [150000, 3000], (-1.0, 128.5), (-1.0, 48.5), [0.0, 1.0], atol=3e-4, rtol=2e-5)
    # (c) two bins (32-row tiles) and eleven bins (4-row tiles)
    one(2, 70, 128, [5000, 1], (-2.0, 129.0), (-2.0, 71.0), [0.0, 1.0], atol=ATOL, rtol=0)
    one(11, 40, 192, [5000], (-2.0, 193.0), (-2.0, 41.0), [0.0, 1.0], atol=ATOL, rtol=0)


def test_dsec_raw_matches_oracle_full_sensor():
    """Raw uint16/int64/uint8 columns + rectify map, DSEC sensor size, 2 samples x 3 sub-windows."""
    from openess_amd import hip
    C, H, W, crop, nwin, n_per = 5, 480, 640, 40, 3, 30000
    rmap = synth.rectify_map(H, W)
    outs_ref, xs, ys, ts, ps = [], [], [], [], []
    for b in range(2):
        x, y, t, p = synth.dsec_raw_events(nwin * n_per, H, W, seed=1205 + b)
        xs.append(x); ys.append(y); ts.append(t); ps.append(p)
    maps = dev(np.stack([rmap, synth.rectify_map(H, W, seed=9)]))
    seg_map = torch.tensor([0] * nwin + [1] * nwin, dtype=torch.int32).cuda()
    so = seg(*([n_per] * (2 * nwin)))
    args = (dev(np.concatenate(xs)), dev(np.concatenate(ys)), dev(np.concatenate(ts)), dev(np.concatenate(ps)), maps,
            seg_map, so, C, H, W)
    for cm in (True, False):
        out = hip.voxelize_dsec_raw(*args, crop_rows=crop, count_mode=cm).cpu().numpy()
        for b in range(2):
            ref = oe.dsec_event_tensor(xs[b], ys[b], ts[b], ps[b], maps[b].cpu().numpy(), nwin, C, H, W, crop,
                                       count_mode=cm)
            got = out[b * nwin * C:(b + 1) * nwin * C]
            if cm:
                assert np.array_equal(got, ref)
            else:
                np.testing.assert_allclose(got, ref, rtol=0, atol=ATOL)


def test_dsec_raw_time_span_beyond_int32():
    """Raw path with a window longer than 2^31 us (the int64 -> float64 -> float32 conversion of t - t[0]; shorter windows take
    the int32 conversion, identical by construction) and with unsorted / repeated timestamps inside the window."""
    from openess_amd import hip
    C, H, W, crop, nwin, n_per = 5, 96, 128, 8, 2, 5000
    rng = np.random.default_rng(17)
    rmap = synth.rectify_map(H, W)
    x = rng.integers(0, W, nwin * n_per).astype(np.uint16)
    y = rng.integers(0, H, nwin * n_per).astype(np.uint16)
    p = rng.integers(0, 2, nwin * n_per).astype(np.uint8)
    t = np.sort(rng.integers(0, 3 * 2 ** 31, nwin * n_per)).astype(np.int64) + 7_000_000_000_000
    t[n_per + 10:n_per + 20] = t[n_per + 10]                       # repeated stamps
    t[n_per + 30], t[n_per + 31] = t[n_per + 31], t[n_per + 30]    # a swapped pair (not sorted)
    maps = dev(rmap[None])
    seg_map = torch.zeros(nwin, dtype=torch.int32).cuda()
    so = seg(*([n_per] * nwin))
    for cm in (True, False):
        out = hip.voxelize_dsec_raw(dev(x), dev(y), dev(t), dev(p), maps, seg_map, so, C, H, W, crop_rows=crop,
                                    count_mode=cm).cpu().numpy()
        ref = oe.dsec_event_tensor(x, y, t, p, rmap, nwin, C, H, W, crop, count_mode=cm)
        if cm:
            assert np.array_equal(out, ref)
        else:
            np.testing.assert_allclose(out, ref, rtol=0, atol=ATOL)


def test_trilinear_full_batch_properties():
    """BASELINE size (8 samples x 20 sub-windows x 100k events, 640x480): size-independent properties."""
    from openess_amd import hip
    C, H, W, crop, nwin, n_per, B = 5, 480, 640, 40, 20, 100000, 8
    rng = np.random.default_rng(5)
    N = B * nwin * n_per
    x = torch.from_numpy(rng.uniform(0, W - 1, N).astype(np.float32)).cuda()
    y = torch.from_numpy(rng.uniform(0, H - 1, N).astype(np.float32)).cuda()
    p = torch.from_numpy(rng.integers(0, 2, N).astype(np.float32)).cuda()
    tt = np.sort(rng.uniform(0, 1, (B * nwin, n_per)).astype(np.float32), axis=1)
    tt[:, 0], tt[:, -1] = 0, 1
    t = torch.from_numpy(tt.reshape(-1)).cuda()
    so = seg(*([n_per] * (B * nwin)))
    out = hip.voxelize_trilinear(x, y, p, t, so, C, H, W, crop_rows=crop)
    assert out.shape == (B * nwin * C, H - crop, W)
    # (1) antisymmetry under polarity flip: value -> -value, same indices
    out_neg = hip.voxelize_trilinear(x, y, 1 - p, t, so, C, H, W, crop_rows=crop)
    assert float((out + out_neg).abs().max()) <= 1e-4
    # (2) partition of unity: with all-positive polarity and in-range coordinates, every event deposits
    #     exactly weight 1 in total over the UNCROPPED grid -> per-segment sum == event count
    full = hip.voxelize_trilinear(x, y, torch.ones_like(p), t, so, C, H, W, crop_rows=0)
    sums = full.view(B * nwin, -1).double().sum(1).cpu().numpy()
    np.testing.assert_allclose(sums, n_per, rtol=2e-5)
    # (3) the cropped result is the row-slice of the uncropped one (same kernel, different tiling)
    full_c = hip.voxelize_trilinear(x, y, p, t, so, C, H, W, crop_rows=0)
    assert float((full_c[:, :H - crop] - out).abs().max()) <= 1e-4
    # (4) count mode: integer grid whose total equals the number of in-grid corners (8 per interior event)
    cnt = hip.voxelize_trilinear(x, y, p, t, so, C, H, W, crop_rows=0, count_mode=True)
    assert bool((cnt == cnt.round()).all())
    t_norm = (4 * t.view(B * nwin, n_per))
    n_t = torch.where(t_norm.int() + 1 < C, 2, 1).sum().item()     # t0+1 == C is masked out
    assert int(cnt.double().sum().item()) == 4 * n_t


@pytest.mark.parametrize("bins", [5, 2, 1])
@pytest.mark.parametrize("sp", [0, 1])
def test_nearest_golden(golden_events, bins, sp):
    from openess_amd import hip
    g = golden_events
    H, W = (int(v) for v in g["near_hw"])
    ev = dev(g["near_ev"])
    out = hip.voxelize_nearest(ev, seg(ev.shape[0]), bins, H, W, separate_pol=bool(sp)).cpu().numpy()
    np.testing.assert_allclose(out, g[f"near_out_b{bins}_sp{sp}"], rtol=0, atol=ATOL)
    cnt = hip.voxelize_nearest(ev, seg(ev.shape[0]), bins, H, W, separate_pol=bool(sp), count_mode=True).cpu().numpy()
    assert np.array_equal(cnt, oe.voxelgrid_nearest(g["near_ev"], (H, W), bins, bool(sp), count_mode=True))


def test_nearest_edge_cases(golden_events):
    from openess_amd import hip
    g = golden_events
    H, W = (int(v) for v in g["near_hw"])
    for key, outk, sp in (("near_evf", "near_outf_b5_sp0", False), ("near_ev0", "near_out0_b5_sp0", False),
                          ("near_ev1", "near_out1_b5_sp1", True)):
        ev = dev(g[key])
        out = hip.voxelize_nearest(ev, seg(ev.shape[0]), 5, H, W, separate_pol=sp).cpu().numpy()
        np.testing.assert_allclose(out, g[outk], rtol=0, atol=ATOL)


def test_nearest_ddd17_sample():
    """DDD17-shaped sample: 20 chunks x 32000 events on 260x346, crop 60 rows (config 1 plumbing)."""
    from openess_amd import hip
    H, W, nchunk, n_per = 260, 346, 20, 32000
    ev = synth.ddd17_events(nchunk * n_per, H, W, seed=11)
    so = seg(*([n_per] * nchunk))
    for bins, sp in ((5, False), (2, True)):
        out = hip.voxelize_nearest(dev(ev), so, bins, H, W, crop_rows=60, separate_pol=sp).cpu().numpy()
        ref = oe.ddd17_event_tensor(ev, nchunk, (H, W), bins, sp, crop_rows=60)
        np.testing.assert_allclose(out, ref, rtol=0, atol=ATOL)
        cnt = hip.voxelize_nearest(dev(ev), so, bins, H, W, crop_rows=60, separate_pol=sp, count_mode=True).cpu().numpy()
        assert np.array_equal(cnt, oe.ddd17_event_tensor(ev, nchunk, (H, W), bins, sp, crop_rows=60, count_mode=True))


def test_histogram_and_normalize(golden_events):
    from openess_amd import hip
    g = golden_events
    H, W = (int(v) for v in g["near_hw"])
    ev = dev(g["hist_ev"])
    out = hip.event_histogram(ev, seg(ev.shape[0]), H, W).cpu().numpy()
    assert np.array_equal(out, g["hist_out"])
    n = hip.masked_normalize(dev(g["norm_in"])).cpu().numpy()
    np.testing.assert_allclose(n, g["norm_out"], rtol=2e-5, atol=2e-6)
    z = hip.masked_normalize(torch.zeros(2, 3, 4, device="cuda")).cpu().numpy()
    assert np.array_equal(z, g["norm_zero_out"])
    # slice variant == dense variant on the sliced copy, and == oracle
    x = torch.randn(2, 10, 6, 8, device="cuda") * (torch.rand(2, 10, 6, 8, device="cuda") > 0.6)
    a = hip.masked_normalize_slice(x, 5, 5).cpu().numpy()
    np.testing.assert_allclose(a, oe.masked_normalize(x[:, 5:10].cpu().numpy()), rtol=2e-5, atol=2e-6)


def test_e2vid_voxel_grid_golden(golden_events):
    """a6: e2vid events_to_voxel_grid (inference_utils.py:405-449) and its torch variant (:452-515) on the HIP
    nearest voxelizer vs the reference golden and the oracle; the caller's array must stay untouched."""
    from openess_amd.e2vid.utils import inference_utils as iu
    g = golden_events
    H, W = (int(v) for v in g["near_hw"])
    ev = np.array(g["e2v_ev"], dtype=np.float64, copy=True)
    keep = ev.copy()
    out = iu.events_to_voxel_grid(ev, 5, W, H)
    assert out.dtype == np.float32 and out.shape == (5, H, W)
    np.testing.assert_allclose(out, g["e2v_out"], rtol=0, atol=ATOL)
    assert np.array_equal(ev, keep)
    out_t = iu.events_to_voxel_grid_pytorch(ev, 5, W, H, torch.device("cuda"))
    assert out_t.is_cuda
    np.testing.assert_allclose(out_t.cpu().numpy(), g["e2v_out"], rtol=0, atol=ATOL)
    # single timestamp (deltaT == 0 -> 1.0) and a larger random set against the oracle
    rng = np.random.default_rng(77)
    n = 20000
    ev2 = np.stack([np.sort(rng.integers(0, 50000, n)).astype(np.float64), rng.integers(0, W, n).astype(np.float64),
                    rng.integers(0, H, n).astype(np.float64), rng.integers(0, 2, n).astype(np.float64)], 1)
    np.testing.assert_allclose(iu.events_to_voxel_grid(ev2, 5, W, H), oe.e2vid_voxel_grid(ev2, 5, W, H), rtol=0, atol=ATOL)
    ev3 = ev2[:100].copy(); ev3[:, 0] = 1234.0
    np.testing.assert_allclose(iu.events_to_voxel_grid(ev3, 5, W, H), oe.e2vid_voxel_grid(ev3, 5, W, H), rtol=0, atol=ATOL)
    with pytest.raises(RuntimeError):
        iu.events_to_voxel_grid_pytorch(ev, 5, W, H, torch.device("cpu"))



@pytest.mark.gpu
def test_masked_stats_slices_equal_per_slice_and_track_rewrites():
    """oess_masked_stats_slices_f32 (all sub-window slices in one launch) == oess_masked_stats_slice_f32 per slice, through the
    product wrapper: event_slice_to_nhwc8 over the 4 slices of a tensor equals the per-slice path bit for bit, also after the
    tensor has been rewritten in place (slice 0 always refreshes the cached statistics)."""
    import torch
    from openess_amd import hip
    torch.manual_seed(5)
    B, n, cs, H, W = 2, 4, 5, 24, 40
    ev = torch.randn(B, n * cs, H, W, device="cuda")
    ev[ev.abs() < 0.8] = 0.0
    for rep in range(2):
        outs = [hip.event_slice_to_nhwc8(ev, i * cs, cs).clone() for i in range(n)]
        for i in range(n):
            single = hip.event_slice_to_nhwc8(ev[:, i * cs:(i + 1) * cs].contiguous(), 0, cs)
            assert torch.equal(outs[i], single), (rep, i)
        st = hip.masked_stats_slices(ev, cs).cpu().numpy()
        for i in range(n):
            sl = ev[:, i * cs:(i + 1) * cs].double()
            np.testing.assert_allclose(st[i, :3], [float(sl.sum()), float((sl * sl).sum()), float((sl != 0).sum())], rtol=1e-12)
        ev.view(-1)[::2].mul_(1.7)                                     # in-place rewrite between the two passes
    # the cache serves only an in-order walk: a raw-pointer rewrite (no version bump, same allocation -- what the voxelizer does
    # every batch) followed by a loop that does NOT start at slice 0 must see the new data
    for i in range(n):
        hip.event_slice_to_nhwc8(ev, i * cs, cs)
    hip._lib.check(hip._lib.load().oess_masked_normalize_f32(ev.data_ptr(), ev.data_ptr(), ev.numel(),
                                                              torch.empty(hip._lib.load().oess_masked_stats_doubles(1), dtype=torch.float64,
                                                                          device="cuda").data_ptr(), None), "rewrite")
    for i in (2, 3, 1):
        got = hip.event_slice_to_nhwc8(ev, i * cs, cs)
        assert torch.equal(got, hip.event_slice_to_nhwc8(ev[:, i * cs:(i + 1) * cs].contiguous(), 0, cs)), i
