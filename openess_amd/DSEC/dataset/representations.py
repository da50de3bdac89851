"""Mirror of DSEC/dataset/representations.py: VoxelGrid (:9-54).  `convert` keeps the reference contract
(four 1-D float32 tensors -> C x H x W grid on the inputs' device); `convert_batch` is the batched form the
loader/trainer should call (all sub-windows of a batch in one launch sequence, crop fused)."""
import torch

from ... import hip


class EventRepresentation:
    def convert(self, x, y, pol, time):
        raise NotImplementedError


class VoxelGrid(EventRepresentation):
    def __init__(self, channels: int, height: int, width: int, normalize: bool):
        self.nb_channels, self.height, self.width = channels, height, width
        self.normalize = normalize

    def convert(self, x, y, pol, time):
        assert x.shape == y.shape == pol.shape == time.shape
        assert x.ndim == 1
        dev = pol.device
        gx, gy, gp, gt = (a.cuda().float().contiguous() for a in (x, y, pol, time))
        seg = torch.tensor([0, x.numel()], dtype=torch.int64)
        grid = hip.voxelize_trilinear(gx, gy, gp, gt, seg, self.nb_channels, self.height, self.width)
        if self.normalize:
            # representations.py:45-53 uses the UNBIASED std of the non-zeros (differs from EventPreprocessor);
            # off in every shipped YAML -> plain tensor ops
            mask = grid != 0
            if bool(mask.any()):
                vals = grid[mask]
                mean, std = vals.mean(), vals.std()
                grid[mask] = (vals - mean) / std if bool(std > 0) else vals - mean
        return grid.to(dev)

    def convert_batch(self, x, y, pol, time, seg_offsets, crop_rows=0, out=None):
        return hip.voxelize_trilinear(x, y, pol, time, seg_offsets, self.nb_channels, self.height, self.width,
                                      crop_rows=crop_rows, out=out)
