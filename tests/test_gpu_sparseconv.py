"""The 3x3 convolution backward over active pixels (csrc/odpl_sparseconv.hip, utils/sparseconv.py) against aten::convolution_backward on the
same (mostly zero) gradient: dX, dW, db; sparse and fully dense gradients, image borders, both weight memory formats."""
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol=1e-4, atol_scale=1e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape
    tol = atol_scale * float(b.abs().max()) + 1e-12
    assert torch.allclose(a, b, rtol=rtol, atol=tol), float((a - b).abs().max())


@pytest.mark.parametrize("B,cin,cout,H,W,density,weights_cl", [(2, 64, 64, 50, 70, 0.01, True), (4, 256, 256, 40, 52, 0.03, True), (2, 128, 64, 33, 47, 1.0, False),
                                                             (1, 64, 128, 96, 100, 0.0, True), (3, 64, 64, 60, 60, 0.2, True)])
def test_sparse_backward_equals_dense_backward(B, cin, cout, H, W, density, weights_cl):
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(cin + H)
    x = (torch.randn(B, cin, H, W, generator=g) * 0.5).to(DEV).contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    if weights_cl:
        conv = conv.to(memory_format=torch.channels_last)
    mask = (torch.rand(B, 1, H, W, generator=g) < density).float()
    mask[:, :, 0, 0] = 1.0 if density > 0 else 0.0                        # a corner pixel: taps that leave the image
    gy = (torch.randn(B, cout, H, W, generator=g) * mask).to(DEV).contiguous(memory_format=torch.channels_last)
    old = sc.min_pixels
    sc.min_pixels = 1
    try:
        x1 = x.clone().requires_grad_()
        assert sc.applicable(x1, conv)
        y1 = sc.conv_bias_act(x1, conv, "relu")
        y1.backward(gy)
        got = (x1.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
        conv.zero_grad(set_to_none=True)
        x2 = x.clone().requires_grad_()
        y2 = F.relu(conv(x2))
        y2.backward(gy)
        _close(y1, y2, 1e-5, 1e-6)
        assert got[0].stride() == x.stride() and got[1].stride() == conv.weight.stride()
        _close(got[0], x2.grad); _close(got[1], conv.weight.grad); _close(got[2], conv.bias.grad)
        # run to run: the same bits
        conv.zero_grad(set_to_none=True)
        x3 = x.clone().requires_grad_()
        sc.conv_bias_act(x3, conv, "relu").backward(gy)
        assert torch.equal(x3.grad, got[0]) and torch.equal(conv.weight.grad, got[1])
    finally:
        sc.min_pixels = old


def test_a_layer_whose_gradient_is_dense_is_sent_to_the_library_route():
    """The share of pixels within reach of a gradient is read back asynchronously; once it is seen above `max_density` the layer stops
    qualifying (`applicable` -> False: the caller takes the dense route) until the next re-probe; a sparse layer keeps qualifying."""
    import time
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(2, 64, 40, 40, generator=g)).to(DEV).contiguous(memory_format=torch.channels_last)
    old = sc.min_pixels
    sc.min_pixels = 1
    try:
        for density, expect in ((1.0, False), (0.01, True)):
            conv = nn.Conv2d(64, 64, 3, 1, 1).to(DEV).to(memory_format=torch.channels_last)
            gy = (torch.randn(2, 64, 40, 40, generator=g) * (torch.rand(2, 1, 40, 40, generator=g) < density).float()).to(DEV).contiguous(memory_format=torch.channels_last)
            assert sc.applicable(x, conv)                                      # nothing known yet
            xi = x.clone().requires_grad_()
            sc.conv_bias_act(xi, conv, "relu").backward(gy)
            torch.cuda.synchronize(); time.sleep(0.05)
            assert sc.applicable(x, conv) is expect
    finally:
        sc.min_pixels = old
