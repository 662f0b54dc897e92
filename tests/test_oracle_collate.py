"""oracle/collate.py sanity (CPU): the fixed-point bilinear restatement has the properties cv2.INTER_LINEAR is known for.
PARITY UNPINNED against cv2 itself (not importable here; the reference holds no image fixture) -- see oracle/collate.py."""
import numpy as np

from oracle import collate as oc


def test_identity_and_constant():
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (17, 23, 3), dtype=np.uint8)
    sq = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    assert np.array_equal(oc.resize_linear_u8(sq, 16), sq)                       # scale 1: a copy
    const = np.full((9, 31, 3), 137, np.uint8)
    assert np.array_equal(oc.resize_linear_u8(const, 20), np.full((20, 20, 3), 137, np.uint8))
    out = oc.resize_linear_u8(img, 40)
    assert out.min() >= img.min() and out.max() <= img.max()                     # convex combination


def test_half_size_is_2x2_mean():
    """cv2 routes exact 2x INTER_LINEAR downscales to the 2x2 area mean (a+b+c+d+2)>>2; the fixed-point path agrees."""
    rng = np.random.default_rng(1)
    img = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8).astype(np.int64)
    want = (img[0::2, 0::2] + img[0::2, 1::2] + img[1::2, 0::2] + img[1::2, 1::2] + 2) >> 2
    assert np.array_equal(oc.resize_linear_u8(img.astype(np.uint8), 16), want.astype(np.uint8))


def test_rows_and_flips():
    img = np.zeros((10, 20, 3), np.uint8)
    x, y = oc.collate([img], [np.array([[2.0, 1.0, 6.0, 4.0, 3.0]])], 8, [3])
    assert x.shape == (1, 3, 8, 8) and y.shape == (1, 6)
    np.testing.assert_allclose(y[0], [0, 3, 1 - 8 / 20, 1 - 5 / 10, 6 / 20, 4 / 10], rtol=1e-6)
    x, y = oc.collate([img, img], [np.zeros((0, 5)), np.array([[0, 0, 20, 10, 1.0]])], 8, [0, 0])
    np.testing.assert_allclose(y, [[1, 1, 0, 0, 1, 1]])
