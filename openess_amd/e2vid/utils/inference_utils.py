"""Mirror of the pieces of e2vid/utils/inference_utils.py that are on the training path:
EventPreprocessor (:49-87) and CropParameters (:284-311)."""
from math import ceil, floor

import torch

from ... import hip


def optimal_crop_size(max_size, max_subsample_factor):
    return int(pow(2, max_subsample_factor) * ceil(max_size / pow(2, max_subsample_factor)))


class CropParameters:
    def __init__(self, width, height, num_encoders):
        self.height, self.width, self.num_encoders = height, width, num_encoders
        self.width_crop_size = optimal_crop_size(width, num_encoders)
        self.height_crop_size = optimal_crop_size(height, num_encoders)
        self.padding_top = ceil(0.5 * (self.height_crop_size - height))
        self.padding_bottom = floor(0.5 * (self.height_crop_size - height))
        self.padding_left = ceil(0.5 * (self.width_crop_size - width))
        self.padding_right = floor(0.5 * (self.width_crop_size - width))
        self.needs_pad = any((self.padding_top, self.padding_bottom, self.padding_left, self.padding_right))
        self.pad = torch.nn.ReflectionPad2d((self.padding_left, self.padding_right, self.padding_top, self.padding_bottom))
        self.cx, self.cy = floor(self.width_crop_size / 2), floor(self.height_crop_size / 2)
        self.ix0, self.ix1 = self.cx - floor(width / 2), self.cx + ceil(width / 2)
        self.iy0, self.iy1 = self.cy - floor(height / 2), self.cy + ceil(height / 2)


class EventPreprocessor:
    """Whole-tensor non-zero mean/std normalisation (hot-pixel list empty and flip off in every shipped
    config, e2vid/options/inference_options.py).  `__call__` keeps the reference contract (tensor in,
    tensor out); `slice_to_nhwc8` is the fused form used by ImageReconstructor."""

    def __init__(self, options=None):
        self.no_normalize = bool(getattr(options, 'no_normalize', False))
        if getattr(options, 'hot_pixels_file', None) or getattr(options, 'flip', False):
            raise NotImplementedError("hot-pixel removal / flip are not on the training path")

    def __call__(self, events):
        if self.no_normalize:
            return events
        return hip.masked_normalize(events.contiguous())

    def slice_to_nhwc8(self, events, c0, cs):
        return hip.event_slice_to_nhwc8(events, c0, cs, normalize=not self.no_normalize)


def _e2vid_grid_hip(events, num_bins, width, height, device):
    """Shared body of the two reference entry points below: one HIP launch sequence of the nearest-xy / linear-t
    voxelizer (oess_voxelize_nearest_f64, signed single grid) over the [N x 4] (t, x, y, p) float64 rows."""
    import numpy as np
    import torch
    from ... import hip
    assert events.shape[1] == 4
    assert num_bins > 0 and width > 0 and height > 0
    ev = torch.as_tensor(np.ascontiguousarray(events) if isinstance(events, np.ndarray) else events)
    ev = ev.to(device=device, dtype=torch.float64)
    xytp = ev[:, [1, 2, 0, 3]].contiguous()              # the kernel's row order is (x, y, t, p)
    seg = torch.tensor([0, ev.shape[0]], dtype=torch.int64)
    return hip.voxelize_nearest(xytp, seg, num_bins, height, width, crop_rows=0, separate_pol=False)


def events_to_voxel_grid(events, num_bins, width, height):
    """e2vid/utils/inference_utils.py:405-449 (NumPy in, NumPy out).  [N x 4] rows (timestamp, x, y, polarity);
    polarity 0 counts as -1; bilinear in time, nearest in space, ONE signed grid.  Runs on the HIP voxelizer of the
    current CUDA device.  Difference from the reference, documented: events outside the grid are dropped (the
    reference lets NumPy wrap negative flat indices / raise IndexError), and the caller's array is not mutated
    (the reference rescales column 0 and rewrites polarity 0 -> -1 in place)."""
    import torch
    return _e2vid_grid_hip(events, num_bins, width, height, torch.device('cuda')).cpu().numpy()


def events_to_voxel_grid_pytorch(events, num_bins, width, height, device):
    """e2vid/utils/inference_utils.py:452-515: same grid, returned as a tensor on `device` (must be a GPU: the
    product path has no CPU fallback)."""
    import torch
    device = torch.device(device)
    if device.type != 'cuda':
        raise RuntimeError("events_to_voxel_grid_pytorch: the HIP voxelizer needs a GPU device")
    return _e2vid_grid_hip(events, num_bins, width, height, device)
