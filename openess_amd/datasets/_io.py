"""Small host-side helpers shared by the dataset mirrors: PNG loading, OpenCV-compatible nearest resize, and the two
torchvision colour augmentations the reference applies to float CHW images."""
import numpy as np
import torch


def load_png(path):
    from PIL import Image
    return np.array(Image.open(path))


def load_png_gray(path):
    """cv2.imread(path, 0) (IMREAD_GRAYSCALE): always one 8-bit channel (datasets/ddd17_events_loader.py:131,
    DSEC/dataset/sequence_ov.py:230 read the GROUND-TRUTH label this way).  8-bit gray files -- what both datasets ship --
    pass through untouched; 16-bit gray is scaled >> 8; palette / colour files are expanded and reduced with OpenCV's
    fixed-point BGR2GRAY rule Y = (4899 R + 9617 G + 1868 B + 8192) >> 14 (restated: cv2 is absent, unpinned for those modes)."""
    from PIL import Image
    im = Image.open(path)
    if im.mode == 'L':
        return np.array(im)
    if im.mode in ('I;16', 'I;16B', 'I'):
        return (np.array(im).astype(np.uint32) >> 8).astype(np.uint8)
    if im.mode == '1':
        return np.array(im.convert('L'))
    rgb = np.array(im.convert('RGB')).astype(np.uint32)
    return ((4899 * rgb[..., 0] + 9617 * rgb[..., 1] + 1868 * rgb[..., 2] + 8192) >> 14).astype(np.uint8)


def image_to_chw_float(path):
    """`np.array(Image.open(p))` -> `[frame / 255]` -> float32 CHW (DSEC/dataset/sequence_ov.py:324-328,
    datasets/ddd17_events_loader.py:214-217): float64 division, then one rounding to float32."""
    frame = load_png(path)
    return torch.from_numpy(np.ascontiguousarray((frame / 255).astype(np.float32).transpose(2, 0, 1)))


def resize_nearest_cv2(img, dsize):
    """cv2.resize(img, (w, h), interpolation=cv2.INTER_NEAREST): src index = min(floor(dst * src/dst), src-1) -- NOT the
    pixel-centre rule PIL / torch use.  cv2 itself is not installed in the build image: restated from OpenCV's rule
    (parity unpinned for the DDD17 label / pl / superpixel resize)."""
    w, h = dsize
    H, W = img.shape[:2]
    ys = np.minimum(np.floor(np.arange(h) * (H / h)).astype(np.int64), H - 1)
    xs = np.minimum(np.floor(np.arange(w) * (W / w)).astype(np.int64), W - 1)
    return img[ys][:, xs]


def _blend(a, b, ratio):
    return (ratio * a + (1.0 - ratio) * b).clamp(0, 1.0).to(a.dtype)


def adjust_brightness(img, factor):
    """torchvision.transforms.functional.adjust_brightness for a float tensor image: blend with black, clamp to [0,1]."""
    return _blend(img, torch.zeros_like(img), factor)


def adjust_contrast(img, factor):
    """torchvision.transforms.functional.adjust_contrast (float CHW, 3 channels): blend with the mean of the grayscale image
    (0.2989 R + 0.587 G + 0.114 B), clamp to [0,1]."""
    if img.shape[-3] == 3:
        r, g, b = img.unbind(dim=-3)
        gray = (0.2989 * r + 0.587 * g + 0.114 * b).to(img.dtype).unsqueeze(-3)
    else:
        gray = img
    mean = torch.mean(gray, dim=(-3, -2, -1), keepdim=True)
    return _blend(img, mean, factor)
