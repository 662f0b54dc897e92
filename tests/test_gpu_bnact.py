import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(4, 64, 26, 26), (2, 32, 52, 52), (3, 1024, 13, 13), (1, 12, 5, 7), (2, 2048, 4, 4)])
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
