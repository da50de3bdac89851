"""Mirror of datasets/data_util.py: generate_input_representation (:6-14), generate_event_histogram (:17-35),
normalize_voxel_grid (:38-48), generate_voxel_grid (:51-117).  NumPy in -> NumPy out like the reference
(the arrays round-trip through the GPU); `*_batch` variants stay on the device."""
import numpy as np
import torch

from .. import hip


def _as_dev(events):
    ev = np.ascontiguousarray(events)
    if ev.dtype != np.int64:
        ev = ev.astype(np.float64)
    return torch.from_numpy(ev).cuda()


def generate_input_representation(events, event_representation, shape, nr_temporal_bins=5, separate_pol=True):
    if event_representation == 'histogram':
        return generate_event_histogram(events, shape)
    elif event_representation == 'voxel_grid':
        return generate_voxel_grid(events, shape, nr_temporal_bins, separate_pol)


def generate_event_histogram(events, shape):
    height, width = shape
    ev = torch.from_numpy(np.ascontiguousarray(events).astype(np.int64)).cuda()
    seg = torch.tensor([0, ev.shape[0]], dtype=torch.int64)
    return hip.event_histogram(ev, seg, height, width).cpu().numpy()


def normalize_voxel_grid(events):
    if events.is_cuda:
        return hip.masked_normalize(events.float().contiguous())
    return hip.masked_normalize(events.float().contiguous().cuda()).to(events.device)


def generate_voxel_grid(events, shape, nr_temporal_bins, separate_pol=True):
    height, width = shape
    assert events.shape[1] == 4 and nr_temporal_bins > 0 and width > 0 and height > 0
    if events.shape[0] == 0:
        raise IndexError("empty event array (the reference indexes events[-1], data_util.py:67)")
    ev = _as_dev(events)
    seg = torch.tensor([0, ev.shape[0]], dtype=torch.int64)
    return hip.voxelize_nearest(ev, seg, nr_temporal_bins, height, width, separate_pol=separate_pol).cpu().numpy()


def generate_voxel_grid_batch(events_dev, seg_offsets, shape, nr_temporal_bins, separate_pol=True, crop_rows=0):
    height, width = shape
    return hip.voxelize_nearest(events_dev, seg_offsets, nr_temporal_bins, height, width, crop_rows=crop_rows,
                                separate_pol=separate_pol)
