#!/usr/bin/env python3
"""Golden vectors for SURVEY 8a rows a7 / a8, produced by RUNNING the reference's own code (imported from /root/reference):
  * `EventPreprocessor.__call__`            e2vid/utils/inference_utils.py:70-87
  * `CropParameters.__init__` (+ its pad)   e2vid/utils/inference_utils.py:284-311
  * `ImageReconstructor.update_reconstruction` x 3 recurrent steps   e2vid/image_reconstructor.py:80-123
Only inputs and expected outputs are stored (weights are the seeded fill of tests/synth.py, keyed by parameter name).

Import stubs (absent third-party modules ONLY; the reference files run unmodified): `cv2`, `albumentations`,
`torchvision.transforms` are imported by these two files but not used on this path (filters / augmentation off) -> empty
modules; `CudaTimer` (e2vid/utils/timers.py:10-26) records torch.cuda events, which needs a CUDA device this build
container does not have -> after import, the NAME `CudaTimer` in the two reference modules is rebound to a no-op context
manager (it has no arithmetic role).
Run:  python tests/golden/gen_golden_e2vid_pre.py"""
import contextlib
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from tests.synth import fill_by_name  # noqa: E402

E2VID_CFG = {'num_bins': 5, 'skip_type': 'sum', 'recurrent_block_type': 'convlstm', 'num_encoders': 3,
             'base_num_channels': 32, 'num_residual_blocks': 2, 'use_upsample_conv': False, 'norm': 'BN'}
CROP_CASES = [(640, 440, 3), (352, 200, 3), (346, 260, 3), (44, 30, 3), (100, 77, 3), (96, 64, 3), (33, 17, 2)]


def install_stubs():
    sys.dont_write_bytecode = True
    for name in ("cv2", "albumentations", "torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    if REF not in sys.path:
        sys.path.insert(0, REF)


class _NoTimer(contextlib.AbstractContextManager):
    def __init__(self, *a, **k):
        pass

    def __exit__(self, *a):
        return False


def main():
    install_stubs()
    import e2vid.utils.inference_utils as iu
    import e2vid.image_reconstructor as ir
    from e2vid.model.model import E2VIDRecurrent
    iu.CudaTimer = ir.CudaTimer = _NoTimer
    torch.manual_seed(1205)
    torch.set_num_threads(4)
    rng = np.random.default_rng(1207)
    out = {}
    opts = SimpleNamespace(no_normalize=False, hot_pixels_file=None, flip=False, use_gpu=False, no_recurrent=False, color=False,
                           auto_hdr=False, auto_hdr_median_filter_size=10, Imin=0.0, Imax=1.0, bilateral_filter_sigma=0.0,
                           unsharp_mask_amount=0.0, unsharp_mask_sigma=1.0, display=False, show_events=False, output_folder=None,
                           dataset_name='x', event_display_mode='red-blue', num_bins_to_show=-1, display_border_crop=0,
                           display_wait_time=1)
    # ---- a7: EventPreprocessor on a sparse tensor, a dense one, an all-zero one and a single-non-zero one
    pre = iu.EventPreprocessor(opts)
    cases = {
        "sparse": (rng.normal(0.3, 1.5, (2, 5, 30, 44)) * (rng.uniform(0, 1, (2, 5, 30, 44)) > 0.8)).astype(np.float32),
        "dense": rng.normal(-0.2, 0.7, (1, 5, 16, 24)).astype(np.float32),
        "zeros": np.zeros((1, 5, 8, 8), np.float32),
    }
    one = np.zeros((1, 5, 8, 8), np.float32)
    one[0, 2, 3, 4] = 1.75                      # one non-zero: stddev = sqrt(x^2 - x^2) = 0 -> 0/0 = NaN at that voxel (reference, unguarded)
    cases["single"] = one
    for k, v in cases.items():
        out[f"pre_in_{k}"] = v
        out[f"pre_out_{k}"] = pre(torch.from_numpy(v.copy())).numpy()
    # ---- a8: CropParameters attributes and its ReflectionPad2d on a small tensor
    attrs = ("width_crop_size", "height_crop_size", "padding_top", "padding_bottom", "padding_left", "padding_right", "cx", "cy",
             "ix0", "ix1", "iy0", "iy1")
    out["crop_cases"] = np.array(CROP_CASES, np.int64)
    out["crop_attrs"] = np.array([[getattr(iu.CropParameters(w, h, n), a) for a in attrs] for (w, h, n) in CROP_CASES], np.int64)
    cp = iu.CropParameters(44, 30, 3)
    pad_in = rng.normal(0, 1, (1, 2, 30, 44)).astype(np.float32)
    out["pad_in"], out["pad_out"] = pad_in, cp.pad(torch.from_numpy(pad_in)).numpy()
    # ---- a8: ImageReconstructor.update_reconstruction, 3 recurrent steps at a size that needs padding (30x44 -> 32x48)
    model = E2VIDRecurrent(E2VID_CFG).eval()
    fill_by_name(model, 11)
    B, H, W = 2, 30, 44
    rec = ir.ImageReconstructor(model, H, W, 5, torch.device("cpu"), opts)
    ev = (rng.normal(0, 1, (B, 15, H, W)) * (rng.uniform(0, 1, (B, 15, H, W)) > 0.7)).astype(np.float32)
    out["rec_events"] = ev
    for i in range(3):
        img, states, latent = rec.update_reconstruction(torch.from_numpy(ev[:, 5 * i:5 * i + 5].copy()))
    out["rec_img"] = img.numpy()
    for k, v in latent.items():
        out[f"rec_latent_{k}"] = v.numpy()
    h, c = rec.last_states_for_each_channel['grayscale'][2]          # deepest ConvLSTM: hidden == latent[8]; keep its cell state
    out["rec_state_c_2"] = c.numpy()
    np.savez_compressed(os.path.join(HERE, "e2vid_pre.npz"), **out)
    print("wrote e2vid_pre.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
