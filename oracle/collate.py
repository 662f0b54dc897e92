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
image (`dataset/collate.photometric_luts` builds it host-side; unpinned like the resize).
ShiftScaleRotate (`Pascal.py:217`, p = 0.2) is restated from albumentations 0.5.2 (`functional.shift_scale_rotate` /
`bbox_shift_scale_rotate`, `Compose`'s `filter_bboxes`) and OpenCV 4.4's `getRotationMatrix2D` / `warpAffine` (imgwarp.cpp:
inverse map in double, AB_BITS = 10 fixed-point coordinates rounded to 1/32 pixel, 2^15-scaled bilinear weights,
`(sum + 2^14) >> 15`, BORDER_REFLECT_101) -- equally UNPINNED: neither library can be imported here.
"""
import math

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


def rotation_matrix(angle, scale, dx, dy, S):
    """cv2.getRotationMatrix2D((S/2, S/2), angle, scale) with albumentations' shift added (functional.shift_scale_rotate)."""
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    c = S / 2
    return np.array([[alpha, beta, (1 - alpha) * c - beta * c + dx * S], [-beta, alpha, beta * c + (1 - alpha) * c + dy * S]], dtype=np.float64)


def _reflect101(p, n):
    p = np.asarray(p).copy()
    if n == 1:
        return np.zeros_like(p)
    while True:
        bad = (p < 0) | (p >= n)
        if not bad.any():
            return p
        p = np.where(p < 0, -p, np.where(p >= n, 2 * n - 2 - p, p))


def warp_affine_u8(img, M):
    """cv2.warpAffine(img, M, (w, h), flags=INTER_LINEAR, borderMode=BORDER_REFLECT_101) on uint8 [S, S, c]."""
    S = img.shape[0]
    m = [float(M[0, 0]), float(M[0, 1]), float(M[0, 2]), float(M[1, 0]), float(M[1, 1]), float(M[1, 2])]
    D = m[0] * m[4] - m[1] * m[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = m[4] * D, m[0] * D
    inv = [A11, m[1] * -D, 0.0, m[3] * -D, A22, 0.0]
    inv[2] = -inv[0] * m[2] - inv[1] * m[5]
    inv[5] = -inv[3] * m[2] - inv[4] * m[5]
    xs = np.arange(S, dtype=np.float64)
    adelta = np.rint(inv[0] * xs * 1024.0).astype(np.int64)
    bdelta = np.rint(inv[3] * xs * 1024.0).astype(np.int64)
    X0 = np.rint((inv[1] * xs + inv[2]) * 1024.0).astype(np.int64) + 16          # indexed by the destination ROW
    Y0 = np.rint((inv[4] * xs + inv[5]) * 1024.0).astype(np.int64) + 16
    X = (X0[:, None] + adelta[None, :]) >> 5
    Y = (Y0[:, None] + bdelta[None, :]) >> 5
    sx, sy = np.clip(X >> 5, -32768, 32767), np.clip(Y >> 5, -32768, 32767)
    fx, fy = X & 31, Y & 31
    x0, x1, y0, y1 = _reflect101(sx, S), _reflect101(sx + 1, S), _reflect101(sy, S), _reflect101(sy + 1, S)
    src = img.astype(np.int64)
    w00, w01, w10, w11 = (32 - fx) * (32 - fy) * 32, fx * (32 - fy) * 32, (32 - fx) * fy * 32, fx * fy * 32
    out = (src[y0, x0] * w00[..., None] + src[y0, x1] * w01[..., None] + src[y1, x0] * w10[..., None] + src[y1, x1] * w11[..., None] + (1 << 14)) >> 15
    return np.clip(out, 0, 255).astype(np.uint8)


def collate(images, objects, S, flips, luts=None, ssr=None):
    """images: list of uint8 [h, w, 3]; objects: list of float [n_i, 5] = (x, y, w, h, label) in source pixels (COCO);
    flips: list of ints, bit 0 horizontal, bit 1 vertical; luts: None or uint8 [B, 3, 256] per-channel tables applied after the
    geometric transforms (cv2.LUT semantics: RandomBrightnessContrast / RGBShift); ssr: None or per image None / (angle, scale,
    dx, dy) of ShiftScaleRotate.  -> (x f32 [B, 3, S, S], y f32 [nT, 6])."""
    xs, ys = [], []
    for i, (img, obj, fl) in enumerate(zip(images, objects, flips)):
        h, w, _ = img.shape
        r = resize_linear_u8(img, S)
        if fl & 1:
            r = r[:, ::-1]
        if fl & 2:
            r = r[::-1]
        M = rotation_matrix(*ssr[i], S) if (ssr is not None and ssr[i] is not None) else None
        if M is not None:
            r = warp_affine_u8(np.ascontiguousarray(r), M)
        if luts is not None:
            r = np.stack([np.asarray(luts[i][c])[r[..., c]] for c in range(3)], -1)
        xs.append((r.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))      # Pascal.py:179,185
        o = np.asarray(obj, dtype=np.float64).reshape(-1, 5)
        x0, y0 = o[:, 0] / w, o[:, 1] / h                                              # albumentations' normalised corners
        x1, y1 = (o[:, 0] + o[:, 2]) / w, (o[:, 1] + o[:, 3]) / h
        if fl & 1:
            x0, x1 = 1.0 - x1, 1.0 - x0
        if fl & 2:
            y0, y1 = 1.0 - y1, 1.0 - y0
        keep = np.ones(len(o), dtype=bool)
        if M is not None:                         # bbox_shift_scale_rotate + Compose's filter_bboxes (min_area = min_visibility = 0)
            px = np.stack([x0, x1, x1, x0], 1) * S
            py = np.stack([y0, y0, y1, y1], 1) * S
            tx = (M[0, 0] * px + M[0, 1] * py + M[0, 2]) / S
            ty = (M[1, 0] * px + M[1, 1] * py + M[1, 2]) / S
            nx0, nx1, ny0, ny1 = tx.min(1), tx.max(1), ty.min(1), ty.max(1)
            full = (nx1 - nx0) * (ny1 - ny0)
            x0, x1, y0, y1 = np.clip(nx0, 0, 1), np.clip(nx1, 0, 1), np.clip(ny0, 0, 1), np.clip(ny1, 0, 1)
            clipped = (x1 - x0) * (y1 - y0)
            with np.errstate(divide="ignore", invalid="ignore"):
                keep = (full != 0) & (clipped != 0) & (clipped / full > 0)
        row = np.stack([np.full(len(o), float(i)), o[:, 4], (x0 * S) / S, (y0 * S) / S, ((x1 - x0) * S) / S, ((y1 - y0) * S) / S], 1)
        ys.append(row.astype(np.float32)[keep])                                        # Pascal.py:144-153,182
    return np.stack(xs), (np.concatenate(ys, 0) if ys else np.zeros((0, 6), np.float32))
