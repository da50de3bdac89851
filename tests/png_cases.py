"""Shared PNG test vectors (written by Pillow = the reference's decoder library, and by oracle.png.encode_gray8 for the format
corners Pillow's encoder never produces)."""
import io
import zlib

import numpy as np

from oracle import png as op


def maps(rng, H, W):
    yy, xx = np.mgrid[0:H, 0:W]
    blocks = ((yy * 10 // H) * 10 + (xx * 10 // W)).astype(np.uint8)                  # 10 x 10 superpixel-like blocks
    labels = rng.integers(0, 11, (max(H // 8, 1), max(W // 8, 1))).astype(np.uint8).repeat(8, 0).repeat(8, 1)[:H, :W]
    labels = np.pad(labels, ((0, H - labels.shape[0]), (0, W - labels.shape[1])), constant_values=255)
    noise = rng.integers(0, 256, (H, W)).astype(np.uint8)
    ramp = ((xx * 3 + yy * 5) & 255).astype(np.uint8)
    smooth = (np.clip(128 + 60 * np.sin(xx / 9.0) + 50 * np.cos(yy / 7.0) + rng.normal(0, 3, (H, W)), 0, 255)).astype(np.uint8)
    return {"blocks": blocks, "labels": labels, "noise": noise, "ramp": ramp, "smooth": smooth}


def pillow_bytes(img, mode="L", **kw):
    from PIL import Image
    im = Image.fromarray(img, mode="L")
    if mode == "P":
        im = im.convert("P") if False else Image.fromarray(img, mode="P")
        im.putpalette([v for i in range(256) for v in (i, 255 - i, (i * 7) & 255)])
    buf = io.BytesIO()
    im.save(buf, format="PNG", **kw)
    return buf.getvalue()


def cases(H, W, seed=0):
    """[(name, file bytes, expected uint8 [H, W])]"""
    rng = np.random.default_rng(seed)
    out = []
    for name, img in maps(rng, H, W).items():
        out.append((f"pil_{name}", pillow_bytes(img), img))
        out.append((f"pil_{name}_l1", pillow_bytes(img, compress_level=1), img))
    m = maps(rng, H, W)
    out.append(("pil_noise_stored", pillow_bytes(m["noise"], compress_level=0), m["noise"]))           # stored blocks
    out.append(("pil_labels_opt", pillow_bytes(m["labels"], optimize=True), m["labels"]))
    out.append(("pil_blocks_palette", pillow_bytes(m["blocks"], mode="P"), m["blocks"]))               # indices of a 'P' image
    for ft in range(5):                                                                                # every filter type, every row
        out.append((f"hand_smooth_f{ft}", op.encode_gray8(m["smooth"], filters=ft), m["smooth"]))
    mixed = [int(v) for v in rng.integers(0, 5, H)]
    out.append(("hand_noise_mixed_filters", op.encode_gray8(m["noise"], filters=mixed, level=9), m["noise"]))
    out.append(("hand_labels_fixed_huffman", op.encode_gray8(m["labels"], filters=1, strategy=zlib.Z_FIXED), m["labels"]))
    out.append(("hand_ramp_split_idat", op.encode_gray8(m["ramp"], filters=mixed, idat_split=97), m["ramp"]))
    out.append(("hand_blocks_rle", op.encode_gray8(m["blocks"], filters=2, strategy=zlib.Z_RLE), m["blocks"]))
    out.append(("hand_noise_huffman_only", op.encode_gray8(m["noise"], filters=0, strategy=zlib.Z_HUFFMAN_ONLY), m["noise"]))
    far = np.tile(rng.integers(0, 256, (1, W)).astype(np.uint8), (H, 1))                               # long-distance matches (one row back .. 32 KB)
    far[::7] = rng.integers(0, 256, (len(range(0, H, 7)), W)).astype(np.uint8)
    out.append(("hand_far_matches", op.encode_gray8(far, filters=0, level=9), far))
    return out
