"""Oracle (test infrastructure): event -> voxel representation builders, NumPy restatement.

Follows, line by line in behaviour (not in text), the reference functions cited below.
All arithmetic types are chosen to reproduce the reference's type promotion exactly so
that *voxel indices are bit-exact* and per-event weights are bit-identical; only the
floating-point summation order may differ from an accelerator implementation.
"""
import numpy as np

F32 = np.float32


# --------------------------------------------------------------------------------------
# a1  VoxelGrid.convert            /root/reference/DSEC/dataset/representations.py:15-54
# --------------------------------------------------------------------------------------
def voxelgrid_trilinear(x, y, pol, t, C, H, W, normalize=False, count_mode=False):
    """Tri-linear splat of +-1 events into a C x H x W float32 grid.

    representations.py:24-25  t_norm = (C-1)*(t-t[0])/(t[-1]-t[0])       (all float32)
    representations.py:27-29  x0,y0,t0 = C-style truncation (Tensor.int())
    representations.py:31     value = 2*pol-1
    representations.py:33-43  8 corners, mask, weight, flat index, put_(accumulate=True)
    representations.py:45-53  optional nonzero mean / unbiased-std normalisation

    count_mode=True replaces every weight by 1.0 (integer histogram of hit indices) and is
    used for the bit-exact index check (SURVEY.md 8c G2').
    """
    x = np.ascontiguousarray(x, dtype=F32)
    y = np.ascontiguousarray(y, dtype=F32)
    pol = np.ascontiguousarray(pol, dtype=F32)
    t = np.ascontiguousarray(t, dtype=F32)
    assert x.shape == y.shape == pol.shape == t.shape and x.ndim == 1
    grid = np.zeros(C * H * W, dtype=F32)
    with np.errstate(all="ignore"):
        # torch: python-int * f32 tensor -> f32 ; then f32 / f32
        t_norm = (F32(C - 1) * (t - t[0])) / (t[-1] - t[0])
        x0 = x.astype(np.int32)
        y0 = y.astype(np.int32)
        t0 = t_norm.astype(np.int32)
        value = F32(2) * pol - F32(1)
        for xlim in (x0, x0 + 1):
            for ylim in (y0, y0 + 1):
                for tlim in (t0, t0 + 1):
                    mask = (xlim < W) & (xlim >= 0) & (ylim < H) & (ylim >= 0) & (tlim >= 0) & (tlim < C)
                    # torch promotes int32 tensor (op) f32 tensor -> f32 (NumPy would give f64)
                    w = value * (F32(1) - np.abs(xlim.astype(F32) - x)) \
                              * (F32(1) - np.abs(ylim.astype(F32) - y)) \
                              * (F32(1) - np.abs(tlim.astype(F32) - t_norm))
                    idx = (H * W) * tlim.astype(np.int64) + W * ylim.astype(np.int64) + xlim.astype(np.int64)
                    if count_mode:
                        np.add.at(grid, idx[mask], F32(1))
                    else:
                        np.add.at(grid, idx[mask], w[mask].astype(F32))
    grid = grid.reshape(C, H, W)
    if normalize and not count_mode:
        nz = grid != 0
        n = int(nz.sum())
        if n > 0:
            vals = grid[nz].astype(np.float64)
            mean = vals.mean()
            std = vals.std(ddof=1) if n > 1 else float("nan")   # torch .std() is unbiased
            if std > 0:
                grid[nz] = ((vals - mean) / std).astype(F32)
            else:
                grid[nz] = (vals - mean).astype(F32)
    return grid


# --------------------------------------------------------------------------------------
# a2  Sequence.events_to_voxel_grid / generate_event_tensor / __getitem__ fixed-count
#     /root/reference/DSEC/dataset/sequence_ov.py:154-165, 204-223, 281-307
# --------------------------------------------------------------------------------------
def dsec_time_normalise(t_chunk):
    """sequence_ov.py:155-156: t=(t-t[0]).astype('float32'); t=t/t[-1]  (t arrives as float64
    because np.stack([x_rect,y_rect,t,p]) promotes, sequence_ov.py:303)."""
    t64 = np.asarray(t_chunk, dtype=np.float64)
    with np.errstate(all="ignore"):
        t = (t64 - t64[0]).astype(F32)
        return t / t[-1]


def dsec_event_tensor(x_raw, y_raw, t_us, p_raw, rectify_map, nr_events_data, C, H, W, crop_rows,
                      count_mode=False):
    """Raw DSEC events of ONE sample -> (nr_events_data*C) x (H-crop_rows) x W float32.

    sequence_ov.py:298-300  rectify_map[y, x] -> (x', y') float32
    sequence_ov.py:302-305  n = N // nr_events_data ; chunk i = events[i*n:(i+1)*n]
    sequence_ov.py:154-165  per-chunk time normalisation, float32 casts
    sequence_ov.py:223      channel slot write
    sequence_ov.py:307      event_tensor[:, :-40, :]
    """
    x_raw = np.asarray(x_raw)
    y_raw = np.asarray(y_raw)
    N = x_raw.shape[0]
    xy = rectify_map[y_raw.astype(np.int64), x_raw.astype(np.int64)]      # N x 2 float32
    xr = xy[:, 0].astype(F32)
    yr = xy[:, 1].astype(F32)
    n = N // nr_events_data
    out = np.zeros((nr_events_data * C, H, W), dtype=F32)
    for i in range(nr_events_data):
        s, e = i * n, (i + 1) * n
        if e <= s:
            continue
        t = dsec_time_normalise(t_us[s:e])
        out[i * C:(i + 1) * C] = voxelgrid_trilinear(xr[s:e], yr[s:e], np.asarray(p_raw[s:e]).astype(F32), t,
                                                     C, H, W, count_mode=count_mode)
    if crop_rows:
        out = out[:, :H - crop_rows, :]
    return out


# --------------------------------------------------------------------------------------
# a3  generate_voxel_grid          /root/reference/datasets/data_util.py:51-117
# --------------------------------------------------------------------------------------
def voxelgrid_nearest(events, shape, nr_temporal_bins, separate_pol=True, count_mode=False):
    """Nearest-xy / linear-t voxel grid, positive and negative polarity accumulated separately.

    events: [N x 4] (x, y, t, p), any numeric dtype (int64 from the DDD17 memmap loader,
    example_loader_ddd17.py:48-52).  NOTE data_util.py:79 mutates the caller's array
    (p==0 -> -1); this restatement works on a copy of the column and returns nothing else.
    """
    height, width = shape
    events = np.asarray(events)
    assert events.shape[1] == 4 and nr_temporal_bins > 0 and width > 0 and height > 0
    pos = np.zeros(nr_temporal_bins * height * width, np.float32)
    neg = np.zeros(nr_temporal_bins * height * width, np.float32)
    if events.shape[0] == 0:
        raise IndexError("empty event array (data_util.py:67 indexes events[-1])")
    last_stamp = events[-1, 2]
    first_stamp = events[0, 2]
    deltaT = last_stamp - first_stamp
    if deltaT == 0:
        deltaT = 1.0
    xs = events[:, 0].astype(np.int64)
    ys = events[:, 1].astype(np.int64)
    ts = (nr_temporal_bins - 1) * (events[:, 2] - first_stamp) / deltaT      # float64
    pols = events[:, 3].copy()
    pols[pols == 0] = -1
    tis = ts.astype(np.int64)
    dts = ts - tis
    vals_left = np.abs(pols) * (1.0 - dts)
    vals_right = np.abs(pols) * dts
    if count_mode:
        vals_left = np.ones_like(vals_left)
        vals_right = np.ones_like(vals_right)
    is_pos = pols == 1
    valid = (xs < width) & (xs >= 0) & (ys < height) & (ys >= 0) & (ts >= 0) & (ts < nr_temporal_bins)
    base = xs + ys * width
    for grid, sel in ((pos, is_pos), (neg, ~is_pos)):
        m = (tis < nr_temporal_bins) & sel & valid
        np.add.at(grid, base[m] + tis[m] * width * height, vals_left[m])
        m = ((tis + 1) < nr_temporal_bins) & sel & valid
        np.add.at(grid, base[m] + (tis[m] + 1) * width * height, vals_right[m])
    pos = pos.reshape(nr_temporal_bins, height, width)
    neg = neg.reshape(nr_temporal_bins, height, width)
    if separate_pol:
        return np.concatenate([pos, neg], axis=0)
    return pos - neg


# --------------------------------------------------------------------------------------
# a4  generate_event_histogram / normalize_voxel_grid   data_util.py:17-35, 38-48
# --------------------------------------------------------------------------------------
def event_histogram(events, shape):
    """2-channel [neg, pos] event-count image; data_util.py:17-35 (no bounds check there:
    out-of-range coordinates raise in NumPy; the restatement keeps that contract)."""
    height, width = shape
    events = np.asarray(events)
    x = events[:, 0].astype(np.int64)
    y = events[:, 1].astype(np.int64)
    p = events[:, 3].copy()
    p[p == 0] = -1
    img_pos = np.zeros(height * width, dtype=np.float32)
    img_neg = np.zeros(height * width, dtype=np.float32)
    np.add.at(img_pos, x[p == 1] + width * y[p == 1], 1)
    np.add.at(img_neg, x[p == -1] + width * y[p == -1], 1)
    return np.stack([img_neg, img_pos], 0).reshape(2, height, width)


def generate_input_representation(events, event_representation, shape, nr_temporal_bins=5, separate_pol=True):
    """Dispatcher, data_util.py:6-14."""
    if event_representation == "histogram":
        return event_histogram(events, shape)
    elif event_representation == "voxel_grid":
        return voxelgrid_nearest(events, shape, nr_temporal_bins, separate_pol)


def masked_normalize(ev):
    """normalize_voxel_grid (data_util.py:38-48) == EventPreprocessor normalisation
    (e2vid/utils/inference_utils.py:78-85): statistics over ALL non-zeros of the whole tensor,
    population variance E[x^2]-mean^2, x <- mask*(x-mean)/std.  float64 accumulation here;
    torch sums float32 pair-wise, so comparisons use a tolerance."""
    ev = np.asarray(ev, dtype=F32)
    nz = ev != 0
    n = int(nz.sum())
    if n == 0:
        return ev.copy()
    s = float(ev.astype(np.float64).sum())
    s2 = float((ev.astype(np.float64) ** 2).sum())
    mean = s / n
    with np.errstate(all="ignore"):
        std = np.sqrt(s2 / n - mean * mean)
        return (nz.astype(F32) * ((ev - F32(mean)) / F32(std))).astype(F32)


# --------------------------------------------------------------------------------------
# a6  e2vid events_to_voxel_grid   /root/reference/e2vid/utils/inference_utils.py:405-449
# --------------------------------------------------------------------------------------
def e2vid_voxel_grid(events, num_bins, width, height):
    """Signed single grid, columns (t, x, y, p); NO spatial bounds check (NumPy add.at raises
    / wraps on bad coordinates exactly like the reference); only `tis < num_bins` tests."""
    events = np.array(events, dtype=np.float64, copy=True)
    assert events.shape[1] == 4
    grid = np.zeros(num_bins * height * width, np.float32)
    last_stamp = events[-1, 0]
    first_stamp = events[0, 0]
    deltaT = last_stamp - first_stamp
    if deltaT == 0:
        deltaT = 1.0
    ts = (num_bins - 1) * (events[:, 0] - first_stamp) / deltaT
    xs = events[:, 1].astype(np.int64)
    ys = events[:, 2].astype(np.int64)
    pols = events[:, 3].copy()
    pols[pols == 0] = -1
    tis = ts.astype(np.int64)
    dts = ts - tis
    vals_left = pols * (1.0 - dts)
    vals_right = pols * dts
    m = tis < num_bins
    np.add.at(grid, xs[m] + ys[m] * width + tis[m] * width * height, vals_left[m])
    m = (tis + 1) < num_bins
    np.add.at(grid, xs[m] + ys[m] * width + (tis[m] + 1) * width * height, vals_right[m])
    return grid.reshape(num_bins, height, width)


# --------------------------------------------------------------------------------------
# a5  DDD17Events.__getitem__ voxel loop   datasets/ddd17_events_loader.py:141-196
# --------------------------------------------------------------------------------------
def ddd17_event_tensor(events, nr_events_data, shape, nr_temporal_bins, separate_pol=False, crop_rows=0,
                       count_mode=False):
    """Fixed-count branch: nr_events_temp = N // nr_events_data; chunk i = [i*n, (i+1)*n)
    (ddd17_events_loader.py:152-165), voxel grid per chunk (:167-173), concat on channels (:191-194),
    crop rows [:-60] (:196).  The bilinear 346->352 resize (:183-189) is a separate resampler and is
    NOT applied here (it is part of the resize op, tested on its own)."""
    events = np.asarray(events)
    N = events.shape[0]
    n = N // nr_events_data
    outs = []
    for i in range(nr_events_data):
        chunk = events[i * n:min((i + 1) * n, N)]
        outs.append(voxelgrid_nearest(chunk, shape, nr_temporal_bins, separate_pol, count_mode=count_mode))
    out = np.concatenate(outs, axis=0)
    if crop_rows:
        out = out[:, :shape[0] - crop_rows, :]
    return out
