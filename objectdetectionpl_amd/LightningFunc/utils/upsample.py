"""Nearest-neighbour 2x upsampling of the YOLO necks (`utils/YoloV4Utils.py:191-206`, `model/YOLOV3.py` nn.Upsample(scale_factor=2),
`utils/YoloV5Utils.py`) on channels-last device tensors: one copy-shaped HIP launch each way instead of ATen's element-per-thread
kernels.  Anything else (CPU, NCHW memory, other sizes, odd channel counts) goes to F.interpolate unchanged."""
import torch
import torch.nn.functional as F

from ... import _lib


def _native(x, size):
    if not (x.is_cuda and x.dim() == 4 and x.dtype in (torch.float32, torch.bfloat16)):
        return False
    B, C, H, W = x.shape
    if tuple(size) != (2 * H, 2 * W) or B * H > 65535 or C % (8 if x.dtype == torch.bfloat16 else 4):
        return False
    return x.is_contiguous(memory_format=torch.channels_last) and not (x.is_contiguous() and C > 1 and H * W > 1)


class _Up2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, C, H, W = x.shape
        y = torch.empty((B, C, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().odpl_upsample2x_nearest(_lib.ptr(x), _lib.ptr(y), B, C, H, W, 1 if x.dtype == torch.bfloat16 else 0,
                                                      C, 0, 0, _lib.current_stream(x.device)), "odpl_upsample2x_nearest")
        ctx.shape = (B, C, H, W)
        return y

    @staticmethod
    def backward(ctx, gy):
        B, C, H, W = ctx.shape
        if not (gy.is_contiguous(memory_format=torch.channels_last) and gy.dtype in (torch.float32, torch.bfloat16)
                and C % (8 if gy.dtype == torch.bfloat16 else 4) == 0):
            gy = gy.contiguous(memory_format=torch.channels_last)
            if gy.dtype not in (torch.float32, torch.bfloat16) or C % (8 if gy.dtype == torch.bfloat16 else 4):
                gy = gy.float()
        gx = torch.empty((B, C, H, W), dtype=gy.dtype, device=gy.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().odpl_upsample2x_nearest(_lib.ptr(gy), _lib.ptr(gx), B, C, H, W, 1 if gy.dtype == torch.bfloat16 else 0,
                                                      C, 0, 1, _lib.current_stream(gy.device)), "odpl_upsample2x_nearest")
        return gx


def upsample_nearest(x, size):
    """F.interpolate(x, size=size, mode="nearest"); the HIP kernels when `size` is exactly twice x's and x is a dense
    channels-last fp32 / bf16 device tensor."""
    size = (int(size[0]), int(size[1]))
    if _native(x, size):
        return _Up2Fn.apply(x)
    return F.interpolate(x, size=size, mode="nearest")


class _UpCatFn(torch.autograd.Function):
    """torch.cat([upsample2x(x), lateral], 1) (or [lateral, upsample2x(x)]) in one buffer: the upsampling kernel writes its channel
    slice directly (pixel stride = all channels), the lateral half is one strided copy; backward reads the upsampled half's gradient
    in place."""

    @staticmethod
    def forward(ctx, x, lateral, up_first):
        B, C, H, W = x.shape
        C2 = lateral.shape[1]
        off_up, off_lat = (0, C) if up_first else (C2, 0)
        out = torch.empty((B, C + C2, 2 * H, 2 * W), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().odpl_upsample2x_nearest(_lib.ptr(x), _lib.ptr(out), B, C, H, W, 1 if x.dtype == torch.bfloat16 else 0,
                                                      C + C2, off_up, 0, _lib.current_stream(x.device)), "odpl_upsample2x_nearest")
        out[:, off_lat:off_lat + C2].copy_(lateral)
        ctx.meta = (B, C, H, W, C2, off_up, off_lat)
        return out

    @staticmethod
    def backward(ctx, g):
        B, C, H, W, C2, off_up, off_lat = ctx.meta
        per = 8 if g.dtype == torch.bfloat16 else 4
        if not (g.is_contiguous(memory_format=torch.channels_last) and g.dtype in (torch.float32, torch.bfloat16)
                and C % per == 0 and C2 % per == 0):
            g = g.float().contiguous(memory_format=torch.channels_last)
        gx = torch.empty((B, C, H, W), dtype=g.dtype, device=g.device, memory_format=torch.channels_last)
        _lib.check(_lib.lib().odpl_upsample2x_nearest(_lib.ptr(g), _lib.ptr(gx), B, C, H, W, 1 if g.dtype == torch.bfloat16 else 0,
                                                      g.shape[1], off_up, 1, _lib.current_stream(g.device)), "odpl_upsample2x_nearest")
        return gx, g[:, off_lat:off_lat + C2], None


cat_enabled = True          # A/B switch (tools/bench_v5.py --no-upsample-cat): False = upsample, then torch.cat


def upsample_cat(x, lateral, up_first=True):
    """torch.cat([F.interpolate(x, scale_factor=2, mode="nearest"), lateral], 1) (YOLOv5 neck) or, with up_first=False,
    torch.cat([lateral, upsampled], 1) (YOLOv4 neck, `utils/YoloV4Utils.py:444-462`) without materialising the upsampled tensor."""
    size = (2 * x.shape[2], 2 * x.shape[3])
    per = 8 if x.dtype == torch.bfloat16 else 4
    if (cat_enabled and _native(x, size) and lateral.dim() == 4 and lateral.dtype == x.dtype and lateral.device == x.device and tuple(lateral.shape[2:]) == size
            and lateral.shape[0] == x.shape[0] and lateral.shape[1] % per == 0):
        return _UpCatFn.apply(x, lateral, bool(up_first))
    up = upsample_nearest(x, size)
    return torch.cat([up, lateral] if up_first else [lateral, up], 1)

