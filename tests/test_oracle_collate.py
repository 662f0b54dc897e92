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


def test_warp_affine_known_answers():
    """ShiftScaleRotate's warp restatement on cases whose answer is known without OpenCV: the identity map reproduces the frame,
    an integer shift moves it and fills the border by reflection-101, a half-pixel shift averages neighbours (fixed point:
    (a + b + 1) >> 1 up to the 2^15 rounding), and boxes follow the forward matrix (hull, clip, drop)."""
    import numpy as np
    from oracle import collate as oc
    rng = np.random.default_rng(0)
    img = rng.integers(0, 256, (12, 12, 3), dtype=np.uint8)
    ident = np.array([[1.0, 0, 0], [0, 1.0, 0]])
    assert np.array_equal(oc.warp_affine_u8(img, ident), img)
    sh = np.array([[1.0, 0, 3.0], [0, 1.0, -2.0]])                 # dst(x, y) = src(x - 3, y + 2)
    out = oc.warp_affine_u8(img, sh)
    assert np.array_equal(out[0:10, 3:12], img[2:12, 0:9])
    assert np.array_equal(out[:, 0], out[:, 6]) and np.array_equal(out[:, 1], out[:, 5])      # reflect-101 about column 3 (src 0)
    half = np.array([[1.0, 0, 0.5], [0, 1.0, 0]])
    out = oc.warp_affine_u8(img, half)
    want = (img[:, :-1].astype(np.int64) * 16384 + img[:, 1:].astype(np.int64) * 16384 + (1 << 14)) >> 15
    assert np.array_equal(out[:, 1:], want.astype(np.uint8))
    # rows: a box pushed out of the frame is dropped, a box half out is clipped
    M = oc.rotation_matrix(0.0, 1.0, 0.0625, 0.0, 64)              # pure shift right by 4 px of a 64-px frame
    imgs = [rng.integers(0, 256, (64, 64, 3), dtype=np.uint8)]
    objs = [np.array([[10, 10, 20, 20, 1], [59, 5, 5, 5, 2], [61, 30, 3, 10, 3]], np.float32)]
    x, y = oc.collate(imgs, objs, 64, [0], ssr=[(0.0, 1.0, 0.0625, 0.0)])
    assert y.shape == (2, 6) and y[:, 1].tolist() == [1.0, 2.0]                               # the third box starts at x = 65 > 64: gone
    assert np.allclose(y[0, 2:], [14 / 64, 10 / 64, 20 / 64, 20 / 64]) and np.allclose(y[1, 2:], [63 / 64, 5 / 64, 1 / 64, 5 / 64])
    assert np.array_equal(x[0], oc.warp_affine_u8(imgs[0], M).astype(np.float32).transpose(2, 0, 1) / np.float32(255.0))
