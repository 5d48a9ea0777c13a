"""ResizeLongestSide.apply_image is Pillow's BILINEAR resize (segment_anything: resize(to_pil_image(image), size)).  The
fixed-point tables and the two integer passes restated in transforms.py - the arithmetic csrc/image.hip resample_u8_kernel runs on the
device - reproduce Pillow bit for bit: enlarging, shrinking (antialias support), odd sizes, one-axis changes, gray and RGB."""
import numpy as np
import pytest
from PIL import Image

from micro_sam_amd import transforms as T


@pytest.mark.parametrize("shape,new", [((96, 128, 3), (768, 1024)), ((300, 200, 3), (1024, 683)), ((513, 700), (751, 1024)),
                                       ((1400, 1100, 3), (1024, 805)), ((64, 64, 3), (64, 200)), ((50, 77), (33, 77)),
                                       ((7, 5, 3), (1024, 731)), ((2000, 37), (1024, 19))])
def test_integer_passes_reproduce_pillow(shape, new):
    rng = np.random.default_rng(sum(shape))
    img = rng.integers(0, 256, size=shape, dtype=np.uint8)
    ref = np.array(Image.fromarray(img).resize((new[1], new[0]), Image.BILINEAR))
    got = T.resize_bilinear_u8_numpy(img, new[0], new[1])
    assert got.shape == ref.shape and np.array_equal(got, ref)


def test_tables():
    b, c = T.pil_bilinear_tables(896, 1024)                   # enlarging: support 1 -> at most 3 taps, coefficients sum to 2^22 (+- rounding)
    assert c.shape == (1024, 3) and b[:, 1].max() <= 3 and b[0].tolist()[0] == 0
    assert np.abs(c.sum(axis=1) - (1 << T.PIL_PRECISION_BITS)).max() <= 2
    b, c = T.pil_bilinear_tables(2048, 1024)                  # shrinking by 2: support 2 -> 5-tap rows
    assert c.shape == (1024, 5) and b[:, 1].max() <= 5
    # the transform class itself is Pillow
    img = np.random.default_rng(0).integers(0, 256, size=(600, 450, 3), dtype=np.uint8)
    out = T.ResizeLongestSide(1024).apply_image(img)
    assert out.shape == (1024, 768, 3) and np.array_equal(out, T.resize_bilinear_u8_numpy(img, 1024, 768))
