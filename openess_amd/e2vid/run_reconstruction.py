#!/usr/bin/env python3
"""Offline E2VID reconstruction (e2vid/run_reconstruction.py:14-116): events text file -> event windows ->
voxel grids (events_to_voxel_grid_pytorch on the HIP voxelizer) -> recurrent E2VID on the MI355X -> one grayscale PNG per
window (what the frame2recon stage reads back as `reconstructions/`).  Same flags as the reference for the path it
executes; `-o/--output_folder` + `--dataset_name` follow e2vid/options/inference_options.py.

    python -m openess_amd.e2vid.run_reconstruction -c E2VID_lightweight.pth.tar -i events.txt -o out/

Checkpoint format: the reference's (`{'arch': 'E2VIDRecurrent', 'model' | 'config.model': {...}, 'state_dict': ...}`,
e2vid/utils/loading_utils.py:5-16); `-c random` builds E2VID_lightweight with seeded random weights (no checkpoint
ships with either repository)."""
import argparse
import os

import numpy as np
import torch

from .image_reconstructor import ImageReconstructor
from .model.model import E2VID_LIGHTWEIGHT_CONFIG, E2VIDRecurrent
from .utils.event_readers import FixedDurationEventReader, FixedSizeEventReader
from .utils.inference_utils import events_to_voxel_grid_pytorch


def load_model(path_to_model):
    if path_to_model == 'random':
        torch.manual_seed(1205)
        return E2VIDRecurrent(E2VID_LIGHTWEIGHT_CONFIG)
    print('Loading model {}...'.format(path_to_model))
    raw = torch.load(path_to_model, map_location='cpu')
    assert raw['arch'] == 'E2VIDRecurrent', raw['arch']
    cfg = raw['model'] if 'model' in raw else raw['config']['model']
    model = E2VIDRecurrent(cfg)
    model.load_state_dict(raw['state_dict'])
    return model


def reconstruct(path_to_events, model, output_folder=None, window_size=None, fixed_duration=False, window_duration=33.33,
                num_events_per_pixel=0.35, skipevents=0, suboffset=0, device='cuda', max_windows=None):
    """Returns the list of reconstructed images (uint8 [H, W]); writes frame_%010d.png + timestamps.txt when a folder is given."""
    with open(path_to_events) as f:
        width, height = (int(v) for v in f.readline().split())
    print('Sensor size: {} x {}'.format(width, height))
    device = torch.device(device)
    model = model.to(device).eval()
    rec = ImageReconstructor(model, height, width, model.num_bins, device)
    N = window_size
    if not fixed_duration and N is None:
        N = int(width * height * num_events_per_pixel)
        print('Will use {} events per tensor (automatically estimated with num_events_per_pixel={:0.2f}).'.format(N, num_events_per_pixel))
    start_index = skipevents + suboffset
    it = (FixedDurationEventReader(path_to_events, duration_ms=window_duration, start_index=start_index) if fixed_duration
          else FixedSizeEventReader(path_to_events, num_events=N, start_index=start_index))
    if output_folder:
        os.makedirs(output_folder, exist_ok=True)
    frames, stamps = [], []
    for k, window in enumerate(it):
        if max_windows is not None and k >= max_windows:
            break
        grid = events_to_voxel_grid_pytorch(window, num_bins=model.num_bins, width=width, height=height, device=device)
        img, _, _ = rec.update_reconstruction(grid.unsqueeze(0), start_index + window.shape[0], window[-1, 0], reconstruct=True)
        if rec.crop.needs_pad:
            img = img[:, :, rec.crop.iy0:rec.crop.iy1, rec.crop.ix0:rec.crop.ix1]
        frame = (img[0, 0].clamp(0, 1) * 255.0).round().to(torch.uint8).cpu().numpy()
        frames.append(frame)
        stamps.append(float(window[-1, 0]))
        if output_folder:
            from PIL import Image
            Image.fromarray(frame).save(os.path.join(output_folder, 'frame_{:010d}.png'.format(k)))
        start_index += window.shape[0]
    if output_folder:
        np.savetxt(os.path.join(output_folder, 'timestamps.txt'), np.asarray(stamps), fmt='%.9f')
    return frames


def main():
    p = argparse.ArgumentParser(description='Evaluating a trained network')
    p.add_argument('-c', '--path_to_model', required=True, type=str)
    p.add_argument('-i', '--input_file', required=True, type=str)
    p.add_argument('--fixed_duration', dest='fixed_duration', action='store_true')
    p.set_defaults(fixed_duration=False)
    p.add_argument('-N', '--window_size', default=None, type=int)
    p.add_argument('-T', '--window_duration', default=33.33, type=float)
    p.add_argument('--num_events_per_pixel', default=0.35, type=float)
    p.add_argument('--skipevents', default=0, type=int)
    p.add_argument('--suboffset', default=0, type=int)
    p.add_argument('--compute_voxel_grid_on_cpu', action='store_true', help='accepted for compatibility; the grid is always built on the GPU')
    p.add_argument('-o', '--output_folder', default=None, type=str)
    p.add_argument('--dataset_name', default='reconstruction', type=str)
    a = p.parse_args()
    out = os.path.join(a.output_folder, a.dataset_name) if a.output_folder else None
    reconstruct(a.input_file, load_model(a.path_to_model), out, a.window_size, a.fixed_duration, a.window_duration,
                a.num_events_per_pixel, a.skipevents, a.suboffset)


if __name__ == "__main__":
    main()
