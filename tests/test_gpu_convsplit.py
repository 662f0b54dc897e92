"""Weight gradients issued on a side stream (LightningFunc/utils/convsplit.py): same numbers as the single-stream autograd path."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_split_backward_matches_plain_conv():
    from objectdetectionpl_amd.LightningFunc.utils import convsplit
    torch.manual_seed(0)
    convs = [torch.nn.Conv2d(16, 32, 3, 1, 1, bias=False).cuda(), torch.nn.Conv2d(32, 32, 1, bias=False).cuda(),
             torch.nn.Conv2d(32, 8, 3, 2, 1, bias=False).cuda()]
    x = torch.randn(4, 16, 26, 26, device="cuda").contiguous(memory_format=torch.channels_last)
    def run(split):
        convsplit.enabled = split
        for c in convs:
            c.zero_grad()
        xi = x.clone().requires_grad_()
        y = xi
        for c in convs:
            y = torch.nn.functional.leaky_relu(convsplit.conv2d(y, c), 0.1)
        y.square().sum().backward()
        torch.cuda.synchronize()
        return [xi.grad.clone()] + [c.weight.grad.clone() for c in convs]
    try:
        ref = run(False)
        for _ in range(3):                      # repeat: allocator reuse across streams must not corrupt anything
            got = run(True)
            for a, b in zip(got, ref):
                assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * float(b.abs().max()))
    finally:
        convsplit.enabled = False
