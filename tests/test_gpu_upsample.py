"""Nearest 2x upsampling (utils/upsample.py, csrc/odpl_resize.hip) and the channels-last RGB Focus fast path against ATen."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(2, 256, 13, 13), (3, 8, 5, 7), (1, 512, 20, 20), (2, 64, 1, 9)])
def test_upsample2x_matches_interpolate(shape, dtype):
    from objectdetectionpl_amd.LightningFunc.utils.upsample import _native, upsample_nearest
    g = torch.Generator().manual_seed(sum(shape))
    x = torch.randn(*shape, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last).requires_grad_()
    size = (2 * shape[2], 2 * shape[3])
    assert _native(x, size)
    y = upsample_nearest(x, size)
    xr = x.detach().clone().requires_grad_()
    yr = F.interpolate(xr, size=size, mode="nearest")
    assert y.shape == yr.shape and y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, yr)
    gy = torch.randn(*y.shape, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    y.backward(gy)
    yr.backward(gy)
    if dtype == torch.float32:
        assert torch.allclose(x.grad, xr.grad, rtol=1e-6, atol=1e-6)           # four addends: association may differ
    else:
        ref = gy.float().view(shape[0], shape[1], shape[2], 2, shape[3], 2).sum((3, 5))
        assert torch.allclose(x.grad.float(), ref, rtol=1e-2, atol=1e-2)        # one bf16 rounding of an fp32 sum
        assert torch.allclose(x.grad.float(), xr.grad.float(), rtol=2e-2, atol=2e-2)


def test_upsample_falls_back_outside_its_domain():
    from objectdetectionpl_amd.LightningFunc.utils.upsample import _native, upsample_nearest
    x = torch.randn(2, 6, 4, 4, device=DEV).contiguous(memory_format=torch.channels_last)       # C % 4 != 0
    assert not _native(x, (8, 8))
    assert torch.equal(upsample_nearest(x, (8, 8)), F.interpolate(x, size=(8, 8), mode="nearest"))
    x = torch.randn(2, 8, 4, 4, device=DEV)                                                     # NCHW memory
    assert not _native(x, (8, 8))
    x = torch.randn(2, 8, 4, 4, device=DEV).contiguous(memory_format=torch.channels_last)
    assert not _native(x, (12, 12))                                                             # not 2x
    assert torch.equal(upsample_nearest(x, (12, 12)), F.interpolate(x, size=(12, 12), mode="nearest"))


def test_upsample_into_a_channel_slice_through_the_c_abi():
    """The large tensor may be a channel slice of a wider channels-last buffer (the concat of the necks)."""
    from objectdetectionpl_amd import _lib
    B, C, H, W, CT, off = 2, 16, 6, 5, 40, 24
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, C, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    big = torch.full((B, CT, 2 * H, 2 * W), -7.0, device=DEV).contiguous(memory_format=torch.channels_last)
    L = _lib.lib()
    dst = big.data_ptr() + off * 4
    _lib.check(L.odpl_upsample2x_nearest(_lib.ptr(x), ctypes.c_void_p(dst), B, C, H, W, 0, CT, 0, 0, _lib.current_stream(torch.device(DEV))), "up")
    torch.cuda.synchronize()
    assert torch.equal(big[:, off:off + C], F.interpolate(x, scale_factor=2, mode="nearest"))
    assert bool((big[:, :off] == -7.0).all()) and bool((big[:, off + C:] == -7.0).all())
    gx = torch.empty_like(x)
    _lib.check(L.odpl_upsample2x_nearest(ctypes.c_void_p(dst), _lib.ptr(gx), B, C, H, W, 0, CT, 0, 1, _lib.current_stream(torch.device(DEV))), "up")
    torch.cuda.synchronize()
    assert torch.allclose(gx, 4 * x, rtol=1e-6, atol=1e-6)
    assert L.odpl_upsample2x_nearest(_lib.ptr(x), ctypes.c_void_p(dst), B, C, H, W, 0, C - 4, 0, 0, None) != 0     # stride < C


@pytest.mark.parametrize("B,H,W", [(2, 8, 12), (3, 64, 64), (1, 2, 2)])
def test_focus_rgb_channels_last_fast_path(B, H, W):
    from objectdetectionpl_amd.LightningFunc.utils.s2d import focus
    g = torch.Generator().manual_seed(B + H)
    x = torch.rand(B, 3, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_()
    y = focus(x)
    ref = torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1)
    assert y.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y, ref)
    gy = torch.randn(*y.shape, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    (gx,) = torch.autograd.grad(y, x, gy)
    (gr,) = torch.autograd.grad(ref, x, gy)
    assert torch.equal(gx, gr)


@pytest.mark.parametrize("up_first", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_upsample_cat_matches_cat_of_interpolate(dtype, up_first):
    """cat([upsample2x(x), lateral], 1) in one buffer (the YOLOv5 neck merges): values and both gradients."""
    from objectdetectionpl_amd.LightningFunc.utils.upsample import upsample_cat
    g = torch.Generator().manual_seed(11)
    B, C, H, W, C2 = 2, 64, 7, 9, 32
    x = torch.randn(B, C, H, W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    lat = torch.randn(B, C2, 2 * H, 2 * W, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    x1, l1, x2, l2 = (t.clone().requires_grad_() for t in (x, lat, x, lat))
    y1 = upsample_cat(x1, l1, up_first)
    u2 = F.interpolate(x2, scale_factor=2, mode="nearest")
    y2 = torch.cat([u2, l2] if up_first else [l2, u2], 1)
    assert y1.shape == y2.shape and y1.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(y1, y2)
    gy = torch.randn(*y2.shape, generator=g).to(DEV).to(dtype).contiguous(memory_format=torch.channels_last)
    y1.backward(gy); y2.backward(gy)
    assert torch.equal(l1.grad, l2.grad)
    tol = dict(rtol=1e-6, atol=1e-6) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(x1.grad.float(), x2.grad.float(), **tol)
