"""Oracle (test infrastructure): ctypes wrapper of the scalar C port (oracle/voxel_oracle.c)."""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libvoxel_oracle.so")
        if not os.path.exists(path):
            raise RuntimeError("oracle C port not built: make -C oracle")
        _LIB = ctypes.CDLL(path)
    return _LIB


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def voxelgrid_trilinear(x, y, p, t, C, H, W, count_mode=False):
    x, y, p, t = (np.ascontiguousarray(a, np.float32) for a in (x, y, p, t))
    grid = np.zeros((C, H, W), np.float32)
    lib().voxelgrid_trilinear_c(_p(x), _p(y), _p(p), _p(t), ctypes.c_int64(x.size), C, H, W, int(count_mode), _p(grid))
    return grid


def dsec_event_tensor(x, y, t, p, rectify_map, nwin, C, H, W, crop, count_mode=False):
    x, y = np.ascontiguousarray(x, np.uint16), np.ascontiguousarray(y, np.uint16)
    t, p = np.ascontiguousarray(t, np.int64), np.ascontiguousarray(p, np.uint8)
    rm = np.ascontiguousarray(rectify_map, np.float32)
    n = x.size // nwin
    out = np.empty((nwin * C, H - crop, W), np.float32)
    scratch = np.empty(C * H * W + 4 * max(n, 1), np.float32)
    lib().dsec_event_tensor_c(_p(x), _p(y), _p(t), _p(p), ctypes.c_int64(x.size), _p(rm), nwin, C, H, W, crop, int(count_mode),
                              _p(out), _p(scratch))
    return out


def voxelgrid_nearest_i64(ev, shape, bins, separate_pol=True, count_mode=False):
    H, W = shape
    ev = np.ascontiguousarray(ev, np.int64)
    out = np.empty(((2 if separate_pol else 1) * bins, H, W), np.float32)
    scratch = np.empty(2 * bins * H * W, np.float32)
    lib().voxelgrid_nearest_i64_c(_p(ev), ctypes.c_int64(ev.shape[0]), bins, H, W, int(separate_pol), int(count_mode), _p(out),
                                  _p(scratch))
    return out
