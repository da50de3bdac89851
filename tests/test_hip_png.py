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


def _rewrap_with_zlib_stream(good, stream):
    """The file `good` with its IDAT payload replaced by `stream`: a VALID chunk container (lengths and CRCs right) around it."""
    import struct
    pos, head, tail = 8, good[:8], b""
    out = [head]
    done = False
    while pos < len(good):
        n, typ = struct.unpack(">I4s", good[pos:pos + 8])
        if typ == b"IDAT":
            if not done:
                out.append(op._chunk(b"IDAT", stream))
                done = True
        else:
            out.append(good[pos:pos + 12 + n])
        pos += 12 + n
    return b"".join(out)


def _idat_stream(good):
    import struct
    pos, parts = 8, []
    while pos < len(good):
        n, typ = struct.unpack(">I4s", good[pos:pos + 8])
        if typ == b"IDAT":
            parts.append(good[pos + 8:pos + 8 + n])
        pos += 12 + n
    return b"".join(parts)


@pytest.mark.timeout(120)
def test_png_truncated_zlib_stream_in_valid_container_terminates_and_is_flagged():
    """A short zlib stream inside an intact IDAT/IEND container: the reader feeds zeros past the end, and for label maps the
    all-zero code of the dynamic table is the most frequent literal, so a decoder without an in-loop bound spins forever.
    The call must return, and every such file comes back either flagged (status 5..8: block / code error, overrun, size) and
    filled with the ignore index, or -- when only the Adler-32 trailer and end-of-block padding were cut, which the decoder does
    not verify -- decoded to exactly the original map.  Never garbage with status 0."""
    H, W = 64, 96
    rng = np.random.default_rng(11)
    m = png_cases.maps(rng, H, W)
    files, names, originals = [], [], []
    for name in ("labels", "blocks", "noise"):
        for kw in ({}, {"strategy": __import__("zlib").Z_FIXED}, {"level": 0}):
            good = op.encode_gray8(m[name], filters=0, **kw)
            z = _idat_stream(good)
            for cut in (len(z) // 4, len(z) // 2, len(z) - 12, len(z) - 5):
                if cut > 8:
                    files.append(_rewrap_with_zlib_stream(good, z[:cut]))
                    names.append((name, tuple(kw), cut, len(z)))
                    originals.append(m[name])
    # a constant map: one literal + one long run per row, the all-zero code is that literal
    const = np.full((H, W), 3, np.uint8)
    good = op.encode_gray8(const, filters=0)
    z = _idat_stream(good)
    for cut in range(6, len(z) - 4, max(1, len(z) // 12)):
        files.append(_rewrap_with_zlib_stream(good, z[:cut]))
        names.append(("const", (), cut, len(z)))
        originals.append(const)
    files.append(good)
    names.append(("const_good", (), len(z), len(z)))
    got, st = _decode(files, H, W)
    assert st[-1] == 0 and np.array_equal(got[-1], const)
    n_flagged = 0
    for i in range(len(files) - 1):
        if st[i] == 0:
            assert np.array_equal(got[i], originals[i]), names[i]          # only the unverified trailer was missing
            assert names[i][2] >= names[i][3] - 12, names[i]                # ... i.e. a cut inside the last bytes of the stream
        else:
            assert st[i] in (5, 6, 7, 8), (names[i], st[i])
            assert (got[i] == 255).all(), names[i]
            n_flagged += 1
    assert n_flagged >= (len(files) - 1) * 2 // 3
