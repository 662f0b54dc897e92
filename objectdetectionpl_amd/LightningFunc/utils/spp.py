"""Fused SPP (three stride-1 max-pools 5/9/13 + concat) of the YOLOv4 neck, `utils/YoloV4Utils.py:400-420`."""
import ctypes

import torch

from ... import _lib


def _strides(t):
    return (ctypes.c_int64 * 4)(*t.stride())


class _SPPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        _lib.require_cuda(x)
        if x.dtype != torch.float32 or x.dim() != 4:
            raise TypeError("spp_cat expects an fp32 [B,C,H,W] tensor")
        B, C, H, W = x.shape
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if not (x.is_contiguous() or cl):
            x = x.contiguous()
        y = torch.empty((B, 4 * C, H, W), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        _lib.check(_lib.lib().odpl_spp_forward(_lib.ptr(x), _strides(x), B, C, H, W, _lib.ptr(y), _strides(y),
                                               _lib.current_stream(x.device)), "odpl_spp_forward")
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        if not (gy.is_contiguous() or gy.is_contiguous(memory_format=torch.channels_last)):
            gy = gy.contiguous()
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().odpl_spp_backward(_lib.ptr(x), _strides(x), B, C, H, W, _lib.ptr(gy.float()), _strides(gy),
                                                _lib.ptr(gx), _lib.current_stream(x.device)), "odpl_spp_backward")
        return gx


def spp_cat(x):
    """cat([maxpool13(x), maxpool9(x), maxpool5(x), x], 1) in one kernel each way (feature maps up to 32 x 32)."""
    return _SPPFn.apply(x)
