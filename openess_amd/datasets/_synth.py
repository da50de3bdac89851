"""Seeded synthetic event / rectify-map generators shaped like the BASELINE configs."""
import numpy as np


def rectify_map(H, W, seed=7):
    """identity + a smooth sub-pixel displacement field (fixed seed) -> [H, W, 2] float32 (x', y')."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    ax, ay, fx, fy = rng.uniform(0.3, 0.9, 4)
    dx = ax * np.sin(2 * np.pi * fx * yy / H + 0.3) * np.cos(2 * np.pi * fy * xx / W) + 0.8 * (xx / W - 0.5)
    dy = ay * np.cos(2 * np.pi * fy * yy / H) * np.sin(2 * np.pi * fx * xx / W + 0.1) - 0.6 * (yy / H - 0.5)
    return np.stack([xx + dx, yy + dy], -1).astype(np.float32)


def dsec_raw_events(n, H, W, seed, span_us=500000):
    """x,y uint16 uniform over the sensor, p uint8 Bernoulli(0.5), t sorted int64 us (ties allowed,
    first/last forced distinct)."""
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n).astype(np.uint16)
    y = rng.integers(0, H, n).astype(np.uint16)
    p = rng.integers(0, 2, n).astype(np.uint8)
    t = np.sort(rng.integers(0, span_us, n)).astype(np.int64) + 1_000_000
    if n > 1 and t[0] == t[-1]:
        t[-1] += 1
    return x, y, t, p


def ddd17_events(n, H, W, seed, span_us=50000):
    rng = np.random.default_rng(seed)
    x = rng.integers(0, W, n)
    y = rng.integers(0, H, n)
    p = rng.integers(0, 2, n)
    t = np.sort(rng.integers(0, span_us, n)) + 5_000_000
    return np.stack([x, y, t, p], -1).astype(np.int64)




def dsec_structured_events(n, H, W, seed, span_us=500000, edge_fraction=0.7, n_edges=200):
    """SURVEY 8d 'structured locality' variant: `edge_fraction` of the events lie on `n_edges` random moving edges (a point
    on a ~40 px segment that translates at up to 0.2 px/ms, +-1 px jitter), the rest uniform -- the tile skew real driving
    scenes produce, for the voxelizer's binning stage."""
    rng = np.random.default_rng(seed)
    t = np.sort(rng.integers(0, span_us, n)).astype(np.int64)
    n_e = int(n * edge_fraction)
    on_edge = np.zeros(n, dtype=bool)
    on_edge[rng.choice(n, n_e, replace=False)] = True
    e = rng.integers(0, n_edges, n)
    x0, y0 = rng.uniform(0, W, n_edges), rng.uniform(0, H, n_edges)
    ang, length = rng.uniform(0, 2 * np.pi, n_edges), rng.uniform(10, 70, n_edges)
    vx, vy = rng.uniform(-0.2e-3, 0.2e-3, n_edges), rng.uniform(-0.2e-3, 0.2e-3, n_edges)      # px per us
    s = rng.uniform(-0.5, 0.5, n)
    xe = x0[e] + s * length[e] * np.cos(ang[e]) + vx[e] * t + rng.normal(0, 1.0, n)
    ye = y0[e] + s * length[e] * np.sin(ang[e]) + vy[e] * t + rng.normal(0, 1.0, n)
    x = np.where(on_edge, np.clip(np.rint(xe), 0, W - 1), rng.integers(0, W, n)).astype(np.uint16)
    y = np.where(on_edge, np.clip(np.rint(ye), 0, H - 1), rng.integers(0, H, n)).astype(np.uint16)
    p = rng.integers(0, 2, n).astype(np.uint8)
    t = t + 1_000_000
    if n > 1 and t[0] == t[-1]:
        t[-1] += 1
    return x, y, t, p
