"""Pin the PNG oracle (oracle/png.py) against Pillow -- the library the reference decodes these files with
(`np.array(Image.open(path))`, DSEC/dataset/sequence_ov.py:343,355) -- and check that the committed test writer produces
files Pillow itself reads back identically.  CPU only."""
import io

import numpy as np
import pytest

from oracle import png as op
from tests import png_cases


@pytest.mark.parametrize("hw", [(23, 37), (64, 96), (440, 640)])
def test_png_oracle_equals_pillow(hw):
    from PIL import Image
    H, W = hw
    for name, data, want in png_cases.cases(H, W, seed=H):
        pil = np.array(Image.open(io.BytesIO(data)))
        assert pil.dtype == np.uint8 and pil.shape == (H, W), name
        assert np.array_equal(pil, want), name                     # the writer's files mean what they should, by Pillow's reading
        got = op.decode_gray8(data)
        assert np.array_equal(got, pil), name
        if name.endswith("labels"):
            assert np.array_equal(op.decode_gray8(data, flip=True), pil[:, ::-1]), name
