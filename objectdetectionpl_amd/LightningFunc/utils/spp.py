"""Fused SPP (three stride-1 max-pools 5/9/13 + concat) of the YOLOv4 neck, `utils/YoloV4Utils.py:400-420`."""
import ctypes

import torch

from ... import _lib


def _strides(t):
    return (ctypes.c_int64 * 4)(*t.stride())


ORDER = {"v4": 0, "v5": 1}


def tile_path(x):
    """True when the channels-last tile kernels apply (the only path that also takes bf16)."""
    B, C, H, W = x.shape
    return (x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous() and
            ((C % 32 == 0 and H * W <= 176) or (C % 16 == 0 and H * W <= 400)))


class _SPPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, order):
        _lib.require_cuda(x)
        if x.dim() != 4 or x.dtype not in (torch.float32, torch.bfloat16) or (x.dtype == torch.bfloat16 and not tile_path(x)):
            raise TypeError("spp_cat expects an fp32 [B,C,H,W] tensor (bf16: dense channels-last, C % 16 == 0, H*W <= 400)")
        B, C, H, W = x.shape
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        if not (x.is_contiguous() or cl):
            x = x.contiguous()
        y = torch.empty((B, 4 * C, H, W), dtype=x.dtype, device=x.device,
                        memory_format=torch.channels_last if cl else torch.contiguous_format)
        _lib.check(_lib.lib().odpl_spp_forward_ex(_lib.ptr(x), _strides(x), B, C, H, W, _lib.ptr(y), _strides(y),
                                                  1 if x.dtype == torch.bfloat16 else 0, order, _lib.current_stream(x.device)), "odpl_spp_forward_ex")
        ctx.save_for_backward(x)
        ctx.order = order
        return y

    @staticmethod
    def backward(ctx, gy):
        (x,) = ctx.saved_tensors
        B, C, H, W = x.shape
        cl = x.is_contiguous(memory_format=torch.channels_last) and not x.is_contiguous()
        gy = gy.to(x.dtype).contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)
        gx = torch.empty_like(x)
        _lib.check(_lib.lib().odpl_spp_backward_ex(_lib.ptr(x), _strides(x), B, C, H, W, _lib.ptr(gy), _strides(gy), _lib.ptr(gx),
                                                   1 if x.dtype == torch.bfloat16 else 0, ctx.order, _lib.current_stream(x.device)),
                   "odpl_spp_backward_ex")
        return gx, None


def spp_cat(x, order="v4"):
    """order "v4": cat([maxpool13(x), maxpool9(x), maxpool5(x), x], 1) (`utils/YoloV4Utils.py:418`);
    order "v5": cat([x, maxpool5(x), maxpool9(x), maxpool13(x)], 1) (`utils/YoloV5Utils.py:110`) -- one kernel each way
    (feature maps up to 32 x 32; bf16 on the channels-last tile path)."""
    return _SPPFn.apply(x, ORDER[order])
