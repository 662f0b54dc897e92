"""CPU oracle for the device-side collate -- TEST INFRASTRUCTURE ONLY.

Restates the train-time input chain of `dataset/Pascal.py:121-186` (== Coco.py:146-220) for the transforms the
device path implements: `A.Resize(S, S)` -> `A.HorizontalFlip` -> `A.VerticalFlip` -> `torch.Tensor(img)` ->
`collate_fn` (`/255`, HWC -> CHW, `data[:, 0] = i`, `cat`).

PARITY UNPINNED: `A.Resize` is `cv2.resize(..., interpolation=cv2.INTER_LINEAR)` on uint8 and neither cv2
(opencv-python, `requirements.txt`) nor albumentations is importable here, and the reference holds no image fixture.
`resize_linear_u8` restates OpenCV's published 8-bit bilinear algorithm (imgproc/resize.cpp: 11-bit fixed-point
coefficients `saturate_cast<short>(c * 2048)`, horizontal pass into int32, vertical pass
`(((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2`, pixel centres `(d + 0.5) * scale - 0.5` evaluated
in double and rounded to float, `scale = 1 / (dst / src)`); the HIP kernel is held bit-exact to THIS restatement.
RandomBrightnessContrast and RGBShift are per-channel cv2.LUT calls in albumentations: the device takes one composed table per
image (`dataset/collate.photometric_luts` builds it host-side; unpinned like the resize).  ShiftScaleRotate (an affine warp,
p = 0.2) is not implemented on the device and not restated.
"""
import numpy as np


def _coeffs(dst, src):
    inv = np.float64(dst) / np.float64(src)
    scale = np.float64(1.0) / inv
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f[lo] = 0.0; s[lo] = 0
    hi = s >= src - 1
    f[hi] = 0.0; s[hi] = src - 1
    a1 = np.rint(f * np.float32(2048.0)).astype(np.int64)                  # cvRound: half to even
    a0 = np.rint((np.float32(1.0) - f) * np.float32(2048.0)).astype(np.int64)
    s1 = np.minimum(s + 1, src - 1)
    return s, s1, a0, a1


def resize_linear_u8(img, S):
    """uint8 [h, w, c] -> uint8 [S, S, c], cv2.INTER_LINEAR semantics (see the module docstring)."""
    h, w, _ = img.shape
    sx, sx1, ax0, ax1 = _coeffs(S, w)
    sy, sy1, ay0, ay1 = _coeffs(S, h)
    src = img.astype(np.int64)
    rows = src[:, sx, :] * ax0[None, :, None] + src[:, sx1, :] * ax1[None, :, None]          # [h, S, c] int
    r0, r1 = rows[sy], rows[sy1]
    out = (((ay0[:, None, None] * (r0 >> 4)) >> 16) + ((ay1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


def collate(images, objects, S, flips, luts=None):
    """images: list of uint8 [h, w, 3]; objects: list of float [n_i, 5] = (x, y, w, h, label) in source pixels (COCO);
    flips: list of ints, bit 0 horizontal, bit 1 vertical; luts: None or uint8 [B, 3, 256] per-channel tables applied to the
    resized frame (cv2.LUT semantics: RandomBrightnessContrast / RGBShift).  -> (x f32 [B, 3, S, S], y f32 [nT, 6])."""
    xs, ys = [], []
    for i, (img, obj, fl) in enumerate(zip(images, objects, flips)):
        h, w, _ = img.shape
        r = resize_linear_u8(img, S)
        if luts is not None:
            r = np.stack([np.asarray(luts[i][c])[r[..., c]] for c in range(3)], -1)
        if fl & 1:
            r = r[:, ::-1]
        if fl & 2:
            r = r[::-1]
        xs.append((r.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))      # Pascal.py:179,185
        o = np.asarray(obj, dtype=np.float64).reshape(-1, 5)
        x0, y0 = o[:, 0] / w, o[:, 1] / h                                              # albumentations' normalised corners
        x1, y1 = (o[:, 0] + o[:, 2]) / w, (o[:, 1] + o[:, 3]) / h
        if fl & 1:
            x0, x1 = 1.0 - x1, 1.0 - x0
        if fl & 2:
            y0, y1 = 1.0 - y1, 1.0 - y0
        row = np.stack([np.full(len(o), float(i)), o[:, 4], (x0 * S) / S, (y0 * S) / S, ((x1 - x0) * S) / S, ((y1 - y0) * S) / S], 1)
        ys.append(row.astype(np.float32))                                              # Pascal.py:144-153,182
    return np.stack(xs), (np.concatenate(ys, 0) if ys else np.zeros((0, 6), np.float32))
