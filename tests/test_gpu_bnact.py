import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(4, 64, 26, 26), (2, 32, 52, 52), (3, 1024, 13, 13), (1, 12, 5, 7), (2, 2048, 4, 4),
                                   (16, 512, 13, 13), (16, 16, 16, 16), (5, 48, 25, 24), (4, 16, 3, 3),
                                   (2, 320, 9, 11), (3, 272, 7, 5), (1, 256, 40, 37)])     # 5 channel slices; 68 quads unsliced; ragged tiles
@pytest.mark.parametrize("act", ["mish", "leaky", "linear", "relu"])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_matches_torch(shape, act, training):
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act, fusable
    g = torch.Generator().manual_seed(shape[1])
    x = (torch.randn(*shape, generator=g) * 1.7 + 0.3).cuda().contiguous(memory_format=torch.channels_last)
    C = shape[1]
    def mk():
        bn = nn.BatchNorm2d(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.2)
            bn.running_mean.copy_(torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(4)) + 0.5)
        return bn.train(training)
    b1, b2 = mk(), mk()
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    assert fusable(x1, b1)
    y1 = bn_act(x1, b1, act)
    z = b2(x2)
    y2 = F.mish(z) if act == "mish" else (F.leaky_relu(z, 0.1) if act == "leaky" else (F.relu(z) if act == "relu" else z))
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-5), float((y1 - y2).abs().max())
    w = torch.randn(y2.shape, generator=torch.Generator().manual_seed(5)).cuda().contiguous(memory_format=torch.channels_last)
    (y1 * w).sum().backward(); (y2 * w).sum().backward()
    sc = float(x2.grad.abs().max())
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-3, atol=2e-5 * sc), float((x1.grad - x2.grad).abs().max())
    assert torch.allclose(b1.weight.grad, b2.weight.grad, rtol=1e-3, atol=1e-4 * float(b2.weight.grad.abs().max()))
    assert torch.allclose(b1.bias.grad, b2.bias.grad, rtol=1e-3, atol=1e-4 * float(b2.bias.grad.abs().max()))
    assert torch.allclose(b1.running_mean, b2.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(b1.running_var, b2.running_var, rtol=1e-5, atol=1e-6)
    assert int(b1.num_batches_tracked) == int(b2.num_batches_tracked)


@pytest.mark.parametrize("shape", [(16, 64, 208, 208), (16, 256, 52, 52), (16, 1024, 13, 13)])
def test_bn_act_at_baseline_config2_sizes(shape):
    """The stage shapes of the BASELINE config-2 step themselves (177 / 44 / 11 MB maps: unsliced, 4 and 16 channel slices, ~1000
    row tiles): statistics against float64 sums, output and gradients against the torch modules."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act
    g = torch.Generator().manual_seed(shape[1])
    C = shape[1]
    x = (torch.randn(*shape, generator=g) * 1.3 - 0.2).cuda().contiguous(memory_format=torch.channels_last)
    b1, b2 = nn.BatchNorm2d(C).cuda().train(), nn.BatchNorm2d(C).cuda().train()
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = bn_act(x1, b1, "mish")
    y2 = F.mish(b2(x2))
    assert torch.allclose(y1, y2, rtol=1e-4, atol=2e-5), float((y1 - y2).abs().max())
    mean64 = x.double().mean((0, 2, 3))
    var64 = x.double().var((0, 2, 3), unbiased=True)
    assert torch.allclose(b1.running_mean.double(), 0.1 * mean64, rtol=1e-6, atol=1e-7)
    assert torch.allclose(b1.running_var.double(), 0.9 + 0.1 * var64, rtol=1e-6, atol=1e-7)
    w = torch.randn(y2.shape, generator=torch.Generator().manual_seed(5)).cuda().contiguous(memory_format=torch.channels_last)
    (y1 * w).sum().backward(); (y2 * w).sum().backward()
    sc = float(x2.grad.abs().max())
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-3, atol=2e-5 * sc), float((x1.grad - x2.grad).abs().max())
    assert torch.allclose(b1.weight.grad, b2.weight.grad, rtol=2e-3, atol=2e-4 * float(b2.weight.grad.abs().max()))
    assert torch.allclose(b1.bias.grad, b2.bias.grad, rtol=2e-3, atol=2e-4 * float(b2.bias.grad.abs().max()))


@pytest.mark.parametrize("shape", [(16, 255, 13, 13), (2, 255, 52, 52), (3, 7, 5, 9), (1, 1, 1, 1)])
def test_conv_bias_head(shape):
    """Linear head conv through conv_bias: same output and same weight / bias / input gradients as nn.Conv2d."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import conv_bias
    torch.manual_seed(7)
    B, C, H, W = shape
    conv = torch.nn.Conv2d(8, C, 1, bias=True).cuda()
    x = torch.randn(B, 8, H, W, device="cuda", requires_grad=True)
    w = torch.randn(B, C, H, W, device="cuda")
    y0 = conv(x)
    (y0 * w).sum().backward()
    ref = (y0.detach().clone(), x.grad.clone(), conv.weight.grad.clone(), conv.bias.grad.clone())
    x.grad = None; conv.zero_grad()
    y1 = conv_bias(x, conv)
    (y1 * w).sum().backward()
    assert torch.allclose(y1, ref[0], rtol=1e-6, atol=1e-6)
    assert torch.allclose(x.grad, ref[1], rtol=1e-5, atol=1e-6)
    assert torch.allclose(conv.weight.grad, ref[2], rtol=1e-4, atol=1e-4)
    assert torch.allclose(conv.bias.grad, ref[3], rtol=1e-5, atol=1e-5 * float(ref[3].abs().max()) + 1e-6)
    # the kernel sums in double: compare against the exact sum too
    exact = w.double().sum((0, 2, 3)).float()
    assert torch.allclose(conv.bias.grad, exact, rtol=1e-6, atol=1e-6 * float(exact.abs().max()) + 1e-7)


@pytest.mark.parametrize("act", ["mish", "leaky"])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_with_residual(act, training):
    """y = act(bn(x)) + residual in one write: same outputs / gradients / running statistics as the three torch modules."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act
    torch.manual_seed(11)
    N, C, H, W = 4, 64, 13, 13
    bn0 = nn.BatchNorm2d(C).cuda(); bn1 = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn0.weight.uniform_(0.5, 1.5); bn0.bias.uniform_(-0.5, 0.5); bn0.running_mean.uniform_(-0.2, 0.2); bn0.running_var.uniform_(0.5, 1.5)
    bn1.load_state_dict(bn0.state_dict())
    bn0.train(training); bn1.train(training)
    x0 = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    r0 = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last).requires_grad_()
    x1 = x0.detach().clone().requires_grad_(); r1 = r0.detach().clone().requires_grad_()
    w = torch.randn(N, C, H, W, device="cuda").contiguous(memory_format=torch.channels_last)
    f = F.mish if act == "mish" else (lambda t: F.leaky_relu(t, 0.1))
    y0 = f(bn0(x0)) + r0
    (y0 * w).sum().backward()
    y1 = bn_act(x1, bn1, act, r1)
    (y1 * w).sum().backward()
    assert torch.allclose(y1, y0, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x1.grad, x0.grad, rtol=1e-4, atol=1e-5)
    assert torch.equal(r1.grad, r0.grad)
    assert torch.allclose(bn1.weight.grad, bn0.weight.grad, rtol=1e-4, atol=1e-4)
    assert torch.allclose(bn1.running_var, bn0.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("training", [True, False])
def test_bn_act_cat(training):
    """cat((mish(bn1(x1)), leaky(bn2(x2))), 1) written in place by the two epilogues: outputs, every gradient and the
    running statistics equal the torch composition."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act_cat
    torch.manual_seed(13)
    N, H, W, C1, C2 = 3, 26, 26, 64, 32
    mods = []
    for C in (C1, C2, C1, C2):
        bn = nn.BatchNorm2d(C).cuda()
        mods.append(bn)
    for a, b in ((0, 2), (1, 3)):
        with torch.no_grad():
            mods[a].weight.uniform_(0.5, 1.5); mods[a].bias.uniform_(-0.5, 0.5); mods[a].running_mean.uniform_(-0.2, 0.2); mods[a].running_var.uniform_(0.5, 1.5)
        mods[b].load_state_dict(mods[a].state_dict())
    for m in mods:
        m.train(training)
    cl = dict(memory_format=torch.channels_last)
    x1 = torch.randn(N, C1, H, W, device="cuda").contiguous(**cl).requires_grad_()
    x2 = torch.randn(N, C2, H, W, device="cuda").contiguous(**cl).requires_grad_()
    x1r, x2r = x1.detach().clone().requires_grad_(), x2.detach().clone().requires_grad_()
    w = torch.randn(N, C1 + C2, H, W, device="cuda").contiguous(**cl)
    ref = torch.cat((F.mish(mods[0](x1r)), F.leaky_relu(mods[1](x2r), 0.1)), 1)
    (ref * w).sum().backward()
    y = bn_act_cat(x1, mods[2], "mish", x2, mods[3], "leaky")
    assert y is not None and y.is_contiguous(**cl)
    (y * w).sum().backward()
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5)
    assert torch.allclose(x1.grad, x1r.grad, rtol=1e-4, atol=1e-5) and torch.allclose(x2.grad, x2r.grad, rtol=1e-4, atol=1e-5)
    for a, b in ((0, 2), (1, 3)):
        assert torch.allclose(mods[b].weight.grad, mods[a].weight.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(mods[b].bias.grad, mods[a].bias.grad, rtol=1e-4, atol=1e-4)
        assert torch.allclose(mods[b].running_mean, mods[a].running_mean, rtol=1e-5, atol=1e-6)
        assert torch.equal(mods[b].num_batches_tracked, mods[a].num_batches_tracked)


@pytest.mark.parametrize("act", ["mish", "leaky", "relu"])
def test_bn_act_inference_path(act):
    """eval mode under no_grad: one launch that reads the running buffers itself -- same numbers as the torch modules, and the
    running statistics / num_batches_tracked are left untouched."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act
    torch.manual_seed(21)
    C = 96
    bn = nn.BatchNorm2d(C).cuda()
    with torch.no_grad():
        bn.weight.uniform_(0.5, 1.5); bn.bias.uniform_(-0.5, 0.5); bn.running_mean.uniform_(-0.3, 0.3); bn.running_var.uniform_(0.2, 2.0)
    bn.eval()
    before = (bn.running_mean.clone(), bn.running_var.clone(), int(bn.num_batches_tracked))
    x = torch.randn(3, C, 20, 20, device="cuda").contiguous(memory_format=torch.channels_last)
    r = torch.randn_like(x)
    f = {"mish": F.mish, "leaky": lambda t: F.leaky_relu(t, 0.1), "relu": F.relu}[act]
    with torch.no_grad():
        y = bn_act(x, bn, act)
        yr = bn_act(x, bn, act, r)
        ref = f(bn(x))
    assert torch.allclose(y, ref, rtol=1e-5, atol=1e-5) and torch.allclose(yr, ref + r, rtol=1e-5, atol=1e-5)
    assert torch.equal(bn.running_mean, before[0]) and torch.equal(bn.running_var, before[1]) and int(bn.num_batches_tracked) == before[2]
    xg = x.clone().requires_grad_()          # eval mode WITH a gradient (RetinaNet.freeze_bn): the autograd path, statistics as constants
    bn_act(xg, bn, act).sum().backward()
    xr = x.clone().requires_grad_()
    f(bn(xr)).sum().backward()
    assert torch.allclose(xg.grad, xr.grad, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("act", ["mish", "leaky", "linear"])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_bf16_activations(act, training):
    """bf16 activations (bf16-autocast network, BASELINE config 5), fp32 parameters and statistics: against the fp32 torch
    composition evaluated on the same (bf16-valued) inputs.  Outputs / input gradients are rounded to bf16 once: 2^-8 relative."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act, fusable
    torch.manual_seed(31)
    N, C, H, W = 4, 64, 20, 20
    def mk():
        bn = nn.BatchNorm2d(C).cuda()
        with torch.no_grad():
            bn.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5)
            bn.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.2)
            bn.running_mean.copy_(torch.randn(C, generator=torch.Generator().manual_seed(3)) * 0.1)
            bn.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(4)) + 0.5)
        return bn.train(training)
    b1, b2 = mk(), mk()
    cl = dict(memory_format=torch.channels_last)
    xb = (torch.randn(N, C, H, W, device="cuda") * 1.5 + 0.2).bfloat16().contiguous(**cl)
    rb = torch.randn(N, C, H, W, device="cuda").bfloat16().contiguous(**cl)
    wb = torch.randn(N, C, H, W, device="cuda").bfloat16().contiguous(**cl)
    x1, r1 = xb.clone().requires_grad_(), rb.clone().requires_grad_()
    assert fusable(x1, b1)
    y1 = bn_act(x1, b1, act, r1)
    assert y1.dtype == torch.bfloat16 and y1.is_contiguous(**cl)
    x2, r2 = xb.float().requires_grad_(), rb.float().requires_grad_()
    z = b2(x2)
    y2 = (F.mish(z) if act == "mish" else (F.leaky_relu(z, 0.1) if act == "leaky" else z)) + r2
    assert torch.allclose(y1.float(), y2, rtol=1e-2, atol=1e-2)
    (y1.float() * wb.float()).sum().backward(); (y2 * wb.float()).sum().backward()
    sc = float(x2.grad.abs().max())
    assert x1.grad.dtype == torch.bfloat16
    assert torch.allclose(x1.grad.float(), x2.grad, rtol=2e-2, atol=1e-2 * sc)
    assert torch.allclose(r1.grad.float(), r2.grad, rtol=1e-2, atol=1e-3)
    assert b1.weight.grad.dtype == torch.float32
    assert torch.allclose(b1.weight.grad, b2.weight.grad, rtol=2e-2, atol=2e-2 * float(b2.weight.grad.abs().max()))
    assert torch.allclose(b1.bias.grad, b2.bias.grad, rtol=2e-2, atol=2e-2 * float(b2.bias.grad.abs().max()))
    assert torch.allclose(b1.running_mean, b2.running_mean, rtol=1e-4, atol=1e-5)
    assert torch.allclose(b1.running_var, b2.running_var, rtol=1e-4, atol=1e-5)


def test_bn_act_cat_bf16():
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act_cat
    torch.manual_seed(33)
    cl = dict(memory_format=torch.channels_last)
    bns = [nn.BatchNorm2d(c).cuda().train() for c in (32, 64, 32, 64)]
    bns[2].load_state_dict(bns[0].state_dict()); bns[3].load_state_dict(bns[1].state_dict())
    x1 = torch.randn(2, 32, 13, 13, device="cuda").bfloat16().contiguous(**cl)
    x2 = torch.randn(2, 64, 13, 13, device="cuda").bfloat16().contiguous(**cl)
    y = bn_act_cat(x1, bns[0], "leaky", x2, bns[1], "leaky")
    ref = torch.cat((F.leaky_relu(bns[2](x1.float()), 0.1), F.leaky_relu(bns[3](x2.float()), 0.1)), 1)
    assert y is not None and y.dtype == torch.bfloat16 and torch.allclose(y.float(), ref, rtol=1e-2, atol=1e-2)


def test_conv_bias_bf16_autocast():
    """Head conv under bf16 autocast: bf16 output, fp32 bias gradient from the bf16 NCHW gradient."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import conv_bias
    torch.manual_seed(9)
    conv = torch.nn.Conv2d(16, 255, 1, bias=True).cuda()
    x = torch.randn(4, 16, 20, 20, device="cuda")
    w = torch.randn(4, 255, 20, 20, device="cuda")
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y0 = conv(x)
    (y0.float() * w).sum().backward()
    ref_b = conv.bias.grad.clone(); conv.zero_grad()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        y1 = conv_bias(x, conv)
    assert y1.dtype == torch.bfloat16 and torch.allclose(y1.float(), y0.float(), rtol=2e-2, atol=2e-2)
    (y1.float() * w).sum().backward()
    assert conv.bias.grad.dtype == torch.float32
    assert torch.allclose(conv.bias.grad, ref_b, rtol=2e-2, atol=2e-2 * float(ref_b.abs().max()))


@pytest.mark.parametrize("act", ["relu", "leaky", "linear"])
@pytest.mark.parametrize("shape", [(2, 256, 25, 42), (3, 16, 7, 5), (1, 512, 13, 13)])
def test_conv_bias_act_matches_torch(shape, act):
    """Conv2d(bias) + activation units without BN (RetinaNet heads): output, input / weight / bias gradients vs the torch ops."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import conv_bias_act
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C + H)
    conv1 = nn.Conv2d(C, C, 3, 1, 1).cuda().to(memory_format=torch.channels_last)
    conv2 = nn.Conv2d(C, C, 3, 1, 1).cuda().to(memory_format=torch.channels_last)
    conv2.load_state_dict(conv1.state_dict())
    x = torch.randn(B, C, H, W, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    x1, x2 = x.clone().requires_grad_(), x.clone().requires_grad_()
    y1 = conv_bias_act(x1, conv1, act)
    z = conv2(x2)
    y2 = F.relu(z) if act == "relu" else (F.leaky_relu(z, 0.1) if act == "leaky" else z)
    assert torch.allclose(y1, y2, rtol=1e-5, atol=1e-6), float((y1 - y2).abs().max())
    w = torch.randn(y2.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    (y1 * w).sum().backward(); (y2 * w).sum().backward()
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-4, atol=1e-5 * float(x2.grad.abs().max()))
    assert torch.allclose(conv1.weight.grad, conv2.weight.grad, rtol=1e-4, atol=1e-5 * float(conv2.weight.grad.abs().max()))
    assert torch.allclose(conv1.bias.grad, conv2.bias.grad, rtol=1e-4, atol=1e-5 * float(conv2.bias.grad.abs().max()))


@pytest.mark.parametrize("training", [True, False])
@pytest.mark.parametrize("shape", [(2, 256, 25, 42), (3, 64, 7, 5)])
def test_bn_add_act_matches_torch(shape, training):
    """relu(bn(x) + s), the tail of a ResNet bottleneck: output, both input gradients, BN parameter gradients, running stats."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_add_act
    B, C, H, W = shape
    g = torch.Generator().manual_seed(C)
    x = torch.randn(*shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    s = torch.randn(*shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    b1, b2 = nn.BatchNorm2d(C).cuda().train(training), nn.BatchNorm2d(C).cuda().train(training)
    with torch.no_grad():
        for b in (b1, b2):
            b.weight.copy_(torch.rand(C, generator=torch.Generator().manual_seed(1)) + 0.5)
            b.bias.copy_(torch.randn(C, generator=torch.Generator().manual_seed(2)) * 0.2)
            b.running_var.copy_(torch.rand(C, generator=torch.Generator().manual_seed(4)) + 0.5)
    x1, x2, s1, s2 = x.clone().requires_grad_(), x.clone().requires_grad_(), s.clone().requires_grad_(), s.clone().requires_grad_()
    y1 = bn_add_act(x1, b1, s1, "relu")
    y2 = F.relu(b2(x2) + s2)
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-5), float((y1 - y2).abs().max())
    w = torch.randn(y2.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    (y1 * w).sum().backward(); (y2 * w).sum().backward()
    # cells whose pre-activation is within rounding of zero may fall on either side of the ReLU: compare away from them
    far = ((b2(x2.detach()) + s2.detach()).abs() > 1e-4) if not training else torch.ones_like(y2, dtype=torch.bool)
    assert torch.allclose(s1.grad[far], s2.grad[far], rtol=1e-5, atol=1e-6)
    sc = float(x2.grad.abs().max())
    assert torch.allclose(x1.grad, x2.grad, rtol=1e-3, atol=5e-5 * sc), float((x1.grad - x2.grad).abs().max())
    assert torch.allclose(b1.weight.grad, b2.weight.grad, rtol=1e-3, atol=1e-4 * float(b2.weight.grad.abs().max()))
    assert torch.allclose(b1.bias.grad, b2.bias.grad, rtol=1e-3, atol=1e-4 * float(b2.bias.grad.abs().max()))
    assert torch.allclose(b1.running_mean, b2.running_mean, rtol=1e-5, atol=1e-6)
    assert torch.allclose(b1.running_var, b2.running_var, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("training", [True, False])
def test_bn_act_of_cat_single_module(training, dtype):
    """leaky(bn(cat(y1, y2))) with ONE BatchNorm2d over the concatenated channels (YOLOv5 BottleneckCSP) as two epilogues over parameter
    slices: output, both input gradients, the module's parameter gradients, running statistics and num_batches_tracked (+1, once)."""
    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act_of_cat
    if _lib.torch_ext() is None:
        pytest.skip("C++ binding not built")
    g = torch.Generator().manual_seed(9)
    B, C1, C2, H, W = 3, 32, 32, 12, 10
    a = torch.randn(B, C1, H, W, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    b = torch.randn(B, C2, H, W, generator=g).cuda().to(dtype).contiguous(memory_format=torch.channels_last)
    m1, m2 = nn.BatchNorm2d(C1 + C2).cuda().train(training), nn.BatchNorm2d(C1 + C2).cuda().train(training)
    with torch.no_grad():
        for m in (m1, m2):
            m.weight.copy_(torch.rand(C1 + C2, generator=torch.Generator().manual_seed(1)) + 0.5)
            m.bias.copy_(torch.randn(C1 + C2, generator=torch.Generator().manual_seed(2)) * 0.2)
            m.running_var.copy_(torch.rand(C1 + C2, generator=torch.Generator().manual_seed(4)) + 0.5)
    a1, b1, a2, b2 = (t.clone().requires_grad_() for t in (a, b, a, b))
    y1 = bn_act_of_cat(a1, b1, m1, "leaky")
    assert y1 is not None and y1.shape == (B, C1 + C2, H, W)
    y2 = F.leaky_relu(m2(torch.cat((a2.float(), b2.float()), 1)), 0.1)
    tol = dict(rtol=1e-4, atol=1e-5) if dtype == torch.float32 else dict(rtol=2e-2, atol=2e-2)
    assert torch.allclose(y1.float(), y2, **tol), float((y1.float() - y2).abs().max())
    w = torch.randn(y2.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    (y1.float() * w).sum().backward(); (y2 * w).sum().backward()
    gt = dict(rtol=1e-3, atol=1e-4) if dtype == torch.float32 else dict(rtol=5e-2, atol=5e-2)
    assert torch.allclose(a1.grad.float(), a2.grad.float(), **gt) and torch.allclose(b1.grad.float(), b2.grad.float(), **gt)
    pt = dict(rtol=1e-3, atol=1e-3) if dtype == torch.float32 else dict(rtol=5e-2, atol=2e-1)
    assert torch.allclose(m1.weight.grad, m2.weight.grad, **pt) and torch.allclose(m1.bias.grad, m2.bias.grad, **pt)
    st = dict(rtol=1e-5, atol=1e-6) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-2)
    assert torch.allclose(m1.running_mean, m2.running_mean, **st) and torch.allclose(m1.running_var, m2.running_var, **st)
    assert int(m1.num_batches_tracked) == int(m2.num_batches_tracked)


@pytest.mark.parametrize("act", ["relu", "leaky", "linear"])
@pytest.mark.parametrize("shape", [(2, 64, 9, 7), (3, 256, 12, 12), (2, 1024, 5, 6), (1, 2048, 3, 3)])
def test_bn_add_act_backward_masks_in_the_reduce_pass(shape, act):
    """`odpl_bn_add_act_backward`: act'(y) from the saved output inside the reduce pass (no separate threshold launch) -- every activation
    it serves, unsliced and channel-sliced reduce grids, against torch's graph; the two gradients are exactly consistent with each other
    (d loss / d s is what the dx pass read)."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_add_act
    DEV = "cuda:0"
    g = torch.Generator().manual_seed(3)
    N, C, H, W = shape
    x = torch.randn(N, H, W, C, generator=g).to(DEV).permute(0, 3, 1, 2)
    s = torch.randn(N, H, W, C, generator=g).to(DEV).permute(0, 3, 1, 2)
    gy = torch.randn(N, H, W, C, generator=g).to(DEV).permute(0, 3, 1, 2)
    b0, b1 = nn.BatchNorm2d(C).to(DEV), nn.BatchNorm2d(C).to(DEV)
    with torch.no_grad():
        b0.weight.copy_(torch.rand(C, generator=g) + 0.5); b0.bias.copy_(torch.randn(C, generator=g) * 0.2)
    b1.load_state_dict(b0.state_dict())
    fn = {"relu": torch.relu, "leaky": lambda t: F.leaky_relu(t, 0.1), "linear": lambda t: t}[act]
    x0, s0 = x.clone().requires_grad_(), s.clone().requires_grad_()
    x1, s1 = x.clone().requires_grad_(), s.clone().requires_grad_()
    y0 = fn(b0(x0) + s0)
    y1 = bn_add_act(x1, b1, s1, act)
    assert torch.allclose(y0, y1, rtol=1e-5, atol=1e-5)
    y0.backward(gy); y1.backward(gy)
    assert torch.allclose(s0.grad, s1.grad, rtol=1e-6, atol=1e-7)
    assert torch.allclose(x0.grad, x1.grad, rtol=2e-4, atol=2e-5 * float(x0.grad.abs().max()))
    assert torch.allclose(b0.weight.grad, b1.weight.grad, rtol=2e-4, atol=1e-4 * float(b0.weight.grad.abs().max()))
    assert torch.allclose(b0.bias.grad, b1.bias.grad, rtol=2e-4, atol=1e-4 * float(b0.bias.grad.abs().max()))
