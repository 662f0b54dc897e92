"""Device-side collate: raw uint8 frames in HBM -> the reference's wire format `(x f32 [B,3,S,S] in [0,1], y f32 [nT,6])`
(`dataset/Pascal.py:121-186`): the whole train transform -- A.Resize (cv2 INTER_LINEAR fixed-point), Horizontal/VerticalFlip,
ShiftScaleRotate (cv2.warpAffine, reflect-101), RandomBrightnessContrast + RGBShift (one composed LUT) -- then /255, HWC->CHW
and the COCO-pixel -> normalised row rewrite, in two HIP launches per batch instead of 5 CPU workers."""
import ctypes
import math

import torch

from .. import _lib


def draw_flips(batch_size, generator=None, p_h=0.2, p_v=0.2):
    """The reference's flip probabilities (`Pascal.py:213-214`); host-side draw, one int per image (bit 0 = H, bit 1 = V)."""
    r = torch.rand(batch_size, 2, generator=generator)
    return ((r[:, 0] < p_h).int() + 2 * (r[:, 1] < p_v).int()).tolist()


def photometric_luts(batch_size, generator=None, p_bc=0.2, p_rgb=0.2, brightness_limit=0.2, contrast_limit=0.2, shift_limit=30):
    """Per-image [3, 256] uint8 tables for the two photometric augmentations of the reference's train transform
    (`Pascal.py:216-217`: `A.RandomBrightnessContrast(p=0.2)`, `A.RGBShift(30, 30, 30, p=0.2)`), composed in that order.
    Both are `cv2.LUT` calls on uint8 in albumentations: `clip(arange(256) * alpha + beta * 255)` then `clip(lut + shift_c)`
    (truncated to uint8).  Channel c of the frame takes the c-th shift as the frame is stored (BGR from cv2: the reference applies
    `r_shift` to channel 0 regardless).  Returns a uint8 tensor [B, 3, 256] (identity rows where nothing fired)."""
    base = torch.arange(256, dtype=torch.float32)
    luts = base.to(torch.uint8).repeat(batch_size, 3, 1)
    u = torch.rand(batch_size, 7, generator=generator)
    for b in range(batch_size):
        lut = base.repeat(3, 1)
        if u[b, 0] < p_bc:
            alpha = 1.0 + (float(u[b, 1]) * 2 - 1) * contrast_limit
            beta = (float(u[b, 2]) * 2 - 1) * brightness_limit
            lut = torch.clamp(lut * alpha + beta * 255.0, 0, 255).to(torch.uint8).float()
        if u[b, 3] < p_rgb:
            shift = (u[b, 4:7] * 2 - 1) * shift_limit
            lut = torch.clamp(lut + shift[:, None], 0, 255).to(torch.uint8).float()
        luts[b] = lut.to(torch.uint8)
    return luts


def draw_shift_scale_rotate(batch_size, generator=None, p=0.2, shift_limit=0.0625, scale_limit=0.1, rotate_limit=45.0):
    """`A.ShiftScaleRotate(p=0.2)` with albumentations 0.5.2's defaults (`Pascal.py:217`): per image None, or
    (angle in degrees, scale, dx, dy) drawn uniformly from (-45, 45), (0.9, 1.1), (-0.0625, 0.0625)^2."""
    u = torch.rand(batch_size, 5, generator=generator)
    out = []
    for b in range(batch_size):
        if u[b, 0] < p:
            out.append(((float(u[b, 1]) * 2 - 1) * rotate_limit, 1.0 + (float(u[b, 2]) * 2 - 1) * scale_limit,
                        (float(u[b, 3]) * 2 - 1) * shift_limit, (float(u[b, 4]) * 2 - 1) * shift_limit))
        else:
            out.append(None)
    return out


def affine_matrices(params, S):
    """(angle, scale, dx, dy) -> (forward 2x3, inverse 2x3) in pixels of the S x S frame, exactly as albumentations 0.5.2 +
    OpenCV build them: `cv2.getRotationMatrix2D((S/2, S/2), angle, scale)`, `M[0,2] += dx*S`, `M[1,2] += dy*S`
    (functional.shift_scale_rotate), and cv2.warpAffine's own inversion of that matrix (imgwarp.cpp), all in double."""
    angle, scale, dx, dy = params
    a = angle * math.pi / 180.0
    alpha, beta = math.cos(a) * scale, math.sin(a) * scale
    cx = cy = S / 2
    M = [alpha, beta, (1 - alpha) * cx - beta * cy + dx * S, -beta, alpha, beta * cx + (1 - alpha) * cy + dy * S]
    D = M[0] * M[4] - M[1] * M[3]
    D = 1.0 / D if D != 0 else 0.0
    A11, A22 = M[4] * D, M[0] * D
    inv = [A11, M[1] * -D, 0.0, M[3] * -D, A22, 0.0]
    inv[2] = -inv[0] * M[2] - inv[1] * M[5]
    inv[5] = -inv[3] * M[2] - inv[4] * M[5]
    return M, inv


def gpu_collate(images, objects, img_size, flips=None, channels_last=False, luts=None, ssr=None):
    """images: list of uint8 [h, w, 3] CUDA tensors (contiguous, any sizes); objects: list of float [n_i, 5] CUDA tensors
    `(x, y, w, h, label)` in source pixels; flips: list of ints (None = no flips); luts: uint8 [B, 3, 256] (None = none),
    see `photometric_luts`; ssr: list of None / (angle, scale, dx, dy) per image, see `draw_shift_scale_rotate`
    -> `(x, y)` on the same device.  A batch in which ShiftScaleRotate fired drops the rows whose box left the frame, as
    albumentations does -- the row count then comes from the device (one host sync, only for those batches)."""
    B = len(images)
    if B == 0:
        raise ValueError("empty batch")
    dev = images[0].device
    _lib.require_cuda(images[0])
    flips = [0] * B if flips is None else [int(f) for f in flips]
    descs = (_lib.Image * B)()
    if luts is not None:
        luts = luts.to(dev, non_blocking=True).contiguous()
        if luts.dtype != torch.uint8 or tuple(luts.shape) != (B, 3, 256):
            raise TypeError("luts must be uint8 [B, 3, 256]")
    for i, im in enumerate(images):
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous() or im.device != dev:
            raise TypeError("images must be contiguous uint8 [h, w, 3] tensors on one device")
        descs[i].data, descs[i].h, descs[i].w, descs[i].flip = im.data_ptr(), im.shape[0], im.shape[1], flips[i]
        descs[i].lut = luts.data_ptr() + 768 * i if luts is not None else None
    x = torch.empty((B, 3, img_size, img_size), dtype=torch.float32, device=dev,
                    memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    stream = _lib.current_stream(dev)
    warped = ssr is not None and any(p is not None for p in ssr)
    inv_host, fwd_rows = None, None
    if warped:
        inv_host = (ctypes.c_double * (6 * B))()
        fwd_rows = []
        for i, prm in enumerate(ssr):
            if prm is None:
                fwd_rows.append([0.0] * 7)
                continue
            fwd, inv = affine_matrices(prm, img_size)
            descs[i].reserved = 1
            inv_host[6 * i:6 * i + 6] = inv
            fwd_rows.append([1.0] + fwd)
    _lib.check(_lib.lib().odpl_collate_images_warp(descs, inv_host, B, img_size, _lib.ptr(x), (ctypes.c_int64 * 4)(*x.stride()), stream),
               "odpl_collate_images_warp")
    counts = [int(o.shape[0]) for o in objects]
    nT = sum(counts)
    y = torch.empty((nT, 6), dtype=torch.float32, device=dev)
    if nT:
        obj = torch.cat([o.reshape(-1, 5).float() for o in objects], 0).contiguous()
        row_img = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(counts)).to(dev, non_blocking=True)
        dims = torch.tensor([[im.shape[0], im.shape[1], f] for im, f in zip(images, flips)], dtype=torch.int32).to(dev, non_blocking=True)
        aff = torch.tensor(fwd_rows, dtype=torch.float64).to(dev, non_blocking=True) if warped else None
        keep = torch.empty(nT, dtype=torch.uint8, device=dev) if warped else None
        _lib.check(_lib.lib().odpl_collate_targets_warp(_lib.ptr(obj), _lib.ptr(row_img), _lib.ptr(dims), _lib.ptr(aff), nT, img_size,
                                                        _lib.ptr(y), _lib.ptr(keep), stream), "odpl_collate_targets_warp")
        if warped:
            y = y[keep.bool()]                      # albumentations' filter_bboxes: rows without area left are dropped
    return x, y
