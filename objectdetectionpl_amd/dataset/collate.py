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


def gpu_collate(images, objects, img_size, flips=None, channels_last=False):
    """images: list of uint8 [h, w, 3] CUDA tensors (contiguous, any sizes); objects: list of float [n_i, 5] CUDA tensors
    `(x, y, w, h, label)` in source pixels; flips: list of ints (None = no flips) -> `(x, y)` on the same device."""
    B = len(images)
    if B == 0:
        raise ValueError("empty batch")
    dev = images[0].device
    _lib.require_cuda(images[0])
    flips = [0] * B if flips is None else [int(f) for f in flips]
    descs = (_lib.Image * B)()
    for i, im in enumerate(images):
        if im.dtype != torch.uint8 or im.dim() != 3 or im.shape[2] != 3 or not im.is_contiguous() or im.device != dev:
            raise TypeError("images must be contiguous uint8 [h, w, 3] tensors on one device")
        descs[i].data, descs[i].h, descs[i].w, descs[i].flip = im.data_ptr(), im.shape[0], im.shape[1], flips[i]
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
