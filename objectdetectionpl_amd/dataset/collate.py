"""Device-side collate: raw uint8 frames in HBM -> the reference's wire format `(x f32 [B,3,S,S] in [0,1], y f32 [nT,6])`
(`dataset/Pascal.py:121-186`): A.Resize (cv2 INTER_LINEAR fixed-point) + Horizontal/VerticalFlip + /255 + HWC->CHW and
the COCO-pixel -> normalised row rewrite, in two HIP launches per batch instead of 5 CPU workers."""
import ctypes

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


def gpu_collate(images, objects, img_size, flips=None, channels_last=False, luts=None):
    """images: list of uint8 [h, w, 3] CUDA tensors (contiguous, any sizes); objects: list of float [n_i, 5] CUDA tensors
    `(x, y, w, h, label)` in source pixels; flips: list of ints (None = no flips); luts: uint8 [B, 3, 256] (None = none),
    see `photometric_luts` -> `(x, y)` on the same device."""
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
    _lib.check(_lib.lib().odpl_collate_images(descs, B, img_size, _lib.ptr(x), (ctypes.c_int64 * 4)(*x.stride()), stream),
               "odpl_collate_images")
    counts = [int(o.shape[0]) for o in objects]
    nT = sum(counts)
    y = torch.empty((nT, 6), dtype=torch.float32, device=dev)
    if nT:
        obj = torch.cat([o.reshape(-1, 5).float() for o in objects], 0).contiguous()
        row_img = torch.repeat_interleave(torch.arange(B, dtype=torch.int32), torch.tensor(counts)).to(dev, non_blocking=True)
        dims = torch.tensor([[im.shape[0], im.shape[1], f] for im, f in zip(images, flips)], dtype=torch.int32).to(dev, non_blocking=True)
        _lib.check(_lib.lib().odpl_collate_targets(_lib.ptr(obj), _lib.ptr(row_img), _lib.ptr(dims), nT, img_size, _lib.ptr(y), stream),
                   "odpl_collate_targets")
    return x, y
