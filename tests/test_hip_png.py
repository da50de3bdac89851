"""GPU parity of the batched PNG decode (oess_png_decode_gray8_batch through the C-ABI) against Pillow -- the
reference's decoder -- and the oracle: bit-exact int64 maps, per-image horizontal flips, every filter type, stored / fixed /
dynamic DEFLATE blocks, split IDAT chunks, palette indices, far matches; corrupt or unsupported files are flagged and filled
with the ignore index instead of producing garbage."""
import io

import numpy as np
import pytest
import torch

from oracle import png as op
from tests import png_cases

pytestmark = pytest.mark.gpu


def _decode(datas, H, W, flips=None):
    from openess_amd import hip
    blob = torch.from_numpy(np.frombuffer(b"".join(datas), np.uint8).copy()).cuda()
    out, st = hip.png_decode_gray8_batch(blob, [len(d) for d in datas], H, W, flips)
    return out.cpu().numpy(), st.cpu().numpy()


@pytest.mark.parametrize("hw", [(23, 37), (64, 96), (440, 640)])
def test_png_batch_decode_equals_pillow(hw):
    from PIL import Image
    H, W = hw
    cs = png_cases.cases(H, W, seed=H + 1)
    flips = [i % 3 == 1 for i in range(len(cs))]
    got, st = _decode([c[1] for c in cs], H, W, flips)
    assert got.dtype == np.int64 and got.shape == (len(cs), H, W)
    for i, (name, data, want) in enumerate(cs):
        assert st[i] == 0, (name, st[i])
        pil = np.array(Image.open(io.BytesIO(data))).astype(np.int64)
        ref = pil[:, ::-1] if flips[i] else pil
        assert np.array_equal(got[i], ref), name
        assert np.array_equal(op.decode_gray8(data, flip=flips[i]).astype(np.int64), ref), name


def test_png_decode_flags_bad_files():
    H, W = 32, 48
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (H, W)).astype(np.uint8)
    good = op.encode_gray8(img, filters=4)
    trunc = good[:len(good) // 2]
    notpng = b"JFIF" + good[4:]
    wrong_size = op.encode_gray8(img[:, :40], filters=0)
    from PIL import Image
    buf = io.BytesIO()
    Image.fromarray(np.stack([img] * 3, -1)).save(buf, format="PNG")                  # RGB: unsupported colour type
    rgb = buf.getvalue()
    corrupt = bytearray(good)
    corrupt[60] ^= 0x5a                                                                 # inside the zlib stream
    got, st = _decode([good, trunc, notpng, wrong_size, rgb, bytes(corrupt), good], H, W)
    assert st[0] == 0 and st[6] == 0 and np.array_equal(got[0], img) and np.array_equal(got[6], img)
    assert st[1] != 0 and st[2] == 1 and st[3] == 8 and st[4] == 3
    for i in (1, 2, 3, 4):
        assert (got[i] == 255).all()
    assert st[5] != 0 or not np.array_equal(got[5], img)                                # a flipped stream bit never passes silently
