"""Oracle (test infrastructure, never imported by the product path): CPU restatement of the PNG decode that the reference does
with PIL -- `np.array(Image.open(path))` at DSEC/dataset/sequence_ov.py:343,355 and datasets/ddd17_events_loader.py:233,257 --
for the 8-bit single-channel maps of the hot path.  The algorithm lives in third-party code the reference depends on and does
not vendor: Pillow (pinned `Pillow==9.0.1` by the reference's requirements) on libpng's format rules and zlib's DEFLATE.  Restated
here from the published format (PNG specification, 2nd ed., sections 5, 9 and 11.2; RFC 1950 / RFC 1951): chunk walk, IDAT
concatenation, zlib inflate (Python's `zlib`, the same library Pillow links), the five scanline filters.
Pinned by tests/test_oracle_png.py against Pillow itself -- the reference's own decoder -- on files written by Pillow and by
the hand-rolled writer below (every filter type, stored / fixed / dynamic blocks, split IDAT chunks, palette indices)."""
import struct
import zlib

import numpy as np

SIG = b"\x89PNG\r\n\x1a\n"


def decode_gray8(data, flip=False):
    """bytes of an 8-bit greyscale (or palette: the INDEX map, as np.array(Image.open()) gives for mode 'P') non-interlaced PNG
    -> uint8 [H, W]."""
    assert data[:8] == SIG, "not a PNG"
    pos, idat, hdr = 8, [], None
    while pos + 12 <= len(data):
        n, typ = struct.unpack(">I4s", data[pos:pos + 8])
        body = data[pos + 8:pos + 8 + n]
        if typ == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        elif typ == b"IDAT":
            idat.append(body)
        elif typ == b"IEND":
            break
        pos += 12 + n
    w, h, depth, ctype, comp, filt, inter = hdr
    assert depth == 8 and ctype in (0, 3) and comp == 0 and filt == 0 and inter == 0, hdr
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), np.uint8)
    assert raw.size == h * (w + 1)
    rows = raw.reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    prev = np.zeros(w, np.int32)
    for y in range(h):
        ft, f = int(rows[y, 0]), rows[y, 1:].astype(np.int32)
        if ft == 0:
            cur = f
        elif ft == 2:
            cur = (f + prev) & 255
        elif ft == 1:
            cur = np.cumsum(f) & 255
        elif ft in (3, 4):
            cur = np.empty(w, np.int32)
            left = upleft = 0
            for x in range(w):
                up = int(prev[x])
                if ft == 3:
                    v = f[x] + ((left + up) >> 1)
                else:
                    p = left + up - upleft
                    pa, pb, pc = abs(p - left), abs(p - up), abs(p - upleft)
                    v = f[x] + (left if (pa <= pb and pa <= pc) else (up if pb <= pc else upleft))
                v &= 255
                cur[x] = v
                left, upleft = v, up
        else:
            raise ValueError(f"bad filter type {ft}")
        out[y] = cur
        prev = cur
    return out[:, ::-1].copy() if flip else out


def _chunk(typ, body):
    return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)


def encode_gray8(img, filters=None, level=6, strategy=zlib.Z_DEFAULT_STRATEGY, idat_split=None, palette=False):
    """Test-vector WRITER (not on any decode path): an 8-bit PNG of `img` with a chosen filter type per row (int or sequence),
    zlib level / strategy (Z_FIXED -> fixed-Huffman blocks, level 0 -> stored blocks) and the zlib stream cut into IDAT chunks
    of `idat_split` bytes."""
    img = np.asarray(img, np.uint8)
    h, w = img.shape
    if filters is None:
        filters = 0
    fl = [int(filters)] * h if np.isscalar(filters) else [int(v) for v in filters]
    rows = bytearray()
    prev = np.zeros(w, np.int32)
    for y in range(h):
        cur = img[y].astype(np.int32)
        left = np.concatenate([[0], cur[:-1]])
        upleft = np.concatenate([[0], prev[:-1]])
        ft = fl[y]
        if ft == 0:
            f = cur
        elif ft == 1:
            f = cur - left
        elif ft == 2:
            f = cur - prev
        elif ft == 3:
            f = cur - ((left + prev) >> 1)
        else:
            p = left + prev - upleft
            pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - upleft)
            pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, upleft))
            f = cur - pred
        rows.append(ft)
        rows += (f & 255).astype(np.uint8).tobytes()
        prev = cur
    co = zlib.compressobj(level, zlib.DEFLATED, 15, 9, strategy)
    z = co.compress(bytes(rows)) + co.flush()
    out = SIG + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 3 if palette else 0, 0, 0, 0))
    if palette:
        out += _chunk(b"PLTE", bytes(range(256)) * 3)
    step = len(z) if not idat_split else idat_split
    for i in range(0, len(z), step):
        out += _chunk(b"IDAT", z[i:i + step])
    return out + _chunk(b"IEND", b"")
