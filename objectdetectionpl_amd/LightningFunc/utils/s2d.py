"""Space-to-depth shuffles of the stems: Focus (`utils/YoloV5Utils.py:117-125`) and the YOLOv2 passthrough "reorg"
(`model/YOLOV2.py:147-151`), one HIP kernel each way (pure permutations, bit-exact)."""
import ctypes

import torch

from ... import _lib


def _strides(t):
    return (ctypes.c_int64 * 4)(*t.stride())


def _dense(t):
    if t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last):
        return t
    return t.contiguous()


class _FocusFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels_last_out):
        _lib.require_cuda(x)
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[2] % 2 or x.shape[3] % 2:
            raise TypeError("focus expects an fp32 [B,C,H,W] tensor with even H and W")
        x = _dense(x)
        B, C, H, W = x.shape
        y = torch.empty((B, 4 * C, H // 2, W // 2), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
        _lib.check(_lib.lib().odpl_focus(_lib.ptr(x), _lib.ptr(y), B, C, H, W, _strides(x), _strides(y), 0,
                                         _lib.current_stream(x.device)), "odpl_focus")
        ctx.meta = (x.shape, x.stride())
        return y

    @staticmethod
    def backward(ctx, gy):
        shape, stride = ctx.meta
        gy = _dense(gy)
        gx = torch.empty_strided(shape, stride, dtype=gy.dtype, device=gy.device)
        B, C, H, W = shape
        _lib.check(_lib.lib().odpl_focus(_lib.ptr(gy), _lib.ptr(gx), B, C, H, W, _strides(gx), _strides(gy), 1,
                                         _lib.current_stream(gy.device)), "odpl_focus")
        return gx, None


class _ReorgFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, channels_last_out):
        _lib.require_cuda(x)
        if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] % 4 or x.shape[2] % 2 or x.shape[3] % 2:
            raise TypeError("reorg expects an fp32 [B,C,H,W] tensor with C % 4 == 0 and even H, W")
        x = x.contiguous()                       # the reference's own .contiguous().view(): NCHW memory is part of the definition
        B, C, H, W = x.shape
        y = torch.empty((B, 4 * C, H // 2, W // 2), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if channels_last_out else torch.contiguous_format)
        _lib.check(_lib.lib().odpl_reorg(_lib.ptr(x), _lib.ptr(y), B, C, H, W, _strides(y), 0, _lib.current_stream(x.device)), "odpl_reorg")
        ctx.shape = x.shape
        return y

    @staticmethod
    def backward(ctx, gy):
        gy = _dense(gy)
        B, C, H, W = ctx.shape
        gx = torch.empty(ctx.shape, dtype=gy.dtype, device=gy.device)
        _lib.check(_lib.lib().odpl_reorg(_lib.ptr(gy), _lib.ptr(gx), B, C, H, W, _strides(gy), 1, _lib.current_stream(gy.device)), "odpl_reorg")
        return gx, None


def focus(x, channels_last_out=None):
    """cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)."""
    if channels_last_out is None:
        channels_last_out = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    return _FocusFn.apply(x, bool(channels_last_out))


def reorg(x, channels_last_out=None):
    """x.contiguous().view(B, C//4, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).contiguous().view(B, -1, H//2, W//2)."""
    if channels_last_out is None:
        channels_last_out = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
    return _ReorgFn.apply(x, bool(channels_last_out))
