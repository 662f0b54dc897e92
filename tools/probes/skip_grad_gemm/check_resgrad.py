"""csrc/odpl_resgrad.hip: backward-data of a residual branch's opening 1x1 convolution with the skip gradient added in the GEMM's
epilogue (utils/YoloV4Utils.py:238-262), against aten::convolution_backward + add."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("cin,cout,h,b", [(64, 64, 26, 3), (128, 128, 13, 2), (64, 32, 20, 3), (256, 256, 9, 2), (512, 512, 13, 1)])
def test_backward_data_plus_skip_vs_aten(cin, cout, h, b):
    from objectdetectionpl_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(b, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).cuda()
    gy = torch.randn(b, cout, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    skip = torch.randn(b, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    M, st = b * h * h, _lib.current_stream(x.device)
    for tile in (0, 64, 128):
        if tile == 128 and cin % 128:
            continue
        for add in (skip, None):
            gx = torch.full_like(x, float("nan"))
            _lib.check(L.odpl_conv1x1_backward_data_add(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(add) if add is not None else None, _lib.ptr(gx), M, cin, cout,
                                                        tile, st), "odpl_conv1x1_backward_data_add")
            want = ref + add if add is not None else ref
            assert torch.allclose(gx, want, rtol=1e-4, atol=1e-5 * float(want.abs().max())), (tile, add is None, float((gx - want).abs().max()))
    # run to run: the same bits (a k-ordered MFMA chain, no atomics)
    a1, a2 = torch.empty_like(x), torch.empty_like(x)
    for o in (a1, a2):
        _lib.check(L.odpl_conv1x1_backward_data_add(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(skip), _lib.ptr(o), M, cin, cout, 0, st), "odpl_conv1x1_backward_data_add")
    assert torch.equal(a1, a2)


def test_residual_unit_gradients_equal_the_autograd_sum():
    """residual_unit(a, b, x) = x + b(a(x)): with the skip gradient folded into a's backward-data GEMM, d loss/d x and every parameter's
    gradient match the plain route (autograd's elementwise add) -- and the forward values are the same tensors' worth of bits."""
    from objectdetectionpl_amd.model.YOLOV4 import ConvUnit, ResChain
    torch.manual_seed(3)
    dev = torch.device("cuda:0")
    chain = ResChain(64, 3).to(dev).train().to(memory_format=torch.channels_last)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(4, 64, 26, 26, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    gout = torch.randn(4, 64, 26, 26, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    res = []
    for fused in (True, False):
        ConvUnit.fuse_skip_grad = fused
        try:
            chain.zero_grad(set_to_none=True)
            x = x0.clone().requires_grad_()
            y = chain(x * 1.0)                      # (a non-leaf input, as inside the network)
            (y * gout).sum().backward()
            res.append((y.detach().clone(), x.grad.clone(), [p.grad.clone() for p in chain.parameters()]))
        finally:
            ConvUnit.fuse_skip_grad = True
    (y1, gx1, gp1), (y2, gx2, gp2) = res
    assert torch.equal(y1, y2)
    assert torch.allclose(gx1, gx2, rtol=1e-4, atol=1e-5 * float(gx2.abs().max()))
    for a, b in zip(gp1, gp2):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5 * float(b.abs().max()) + 1e-9)
