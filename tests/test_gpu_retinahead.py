"""RetinaNet's final head convolutions hosted by `utils/retinahead.py` and their backward from the criterion's STRUCTURED gradient
(`csrc/odpl_retinagrad.hip`): same numbers as the dense route (conv -> dense d loss/d preds -> aten convolution_backward, the
reference's implicit autograd of model/RetinaNet.py:101-113 + losses.py:405-472) and as the CPU oracle end to end."""
import math

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from oracle import retina as oracle_retina

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol=1e-4, atol_scale=1e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape
    tol = atol_scale * float(b.abs().max()) + 1e-12
    assert torch.allclose(a, b, rtol=rtol, atol=tol), float((a - b).abs().max())


def _levels(w, h):
    return [(math.ceil(h / 2 ** (3 + i)), math.ceil(w / 2 ** (3 + i))) for i in range(5)]


def _targets(g, B, per, C):
    n = B * per
    wh = torch.rand(n, 2, generator=g) * 0.35 + 0.15
    cxy = torch.rand(n, 2, generator=g) * (1 - wh) + wh / 2
    return torch.cat([torch.arange(B).repeat_interleave(per).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(), cxy, wh], 1)


def _ref_forward(xl, xc, wl, bl, wc, bc, C):
    B = xl[0].shape[0]
    loc = torch.cat([F.conv2d(x, wl, bl, padding=1).permute(0, 2, 3, 1).reshape(B, -1, 4) for x in xl], 1)
    cls = torch.cat([F.conv2d(x, wc, bc, padding=1).permute(0, 2, 3, 1).reshape(B, -1, C) for x in xc], 1)
    return loc, cls


@pytest.mark.parametrize("wh,cin,C,B,weights_cl", [((128, 128), 16, 5, 2, True), ((160, 96), 24, 3, 3, False), ((256, 256), 256, 80, 2, True),
                                                   ((1333, 800), 256, 80, 4, True)])     # the last: BASELINE config 4 at full size (200 700 anchors)
def test_structured_backward_equals_dense_route_and_oracle(wh, cin, C, B, weights_cl):
    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    g = torch.Generator().manual_seed(cin + C)
    levels = _levels(*wh)
    anchors = get_anchor_boxes(torch.Tensor(list(wh)), torch.device(DEV))
    assert anchors.shape[0] == 9 * sum(h * w for h, w in levels)
    xl = [torch.randn(B, cin, h, w, generator=g) * 0.5 for h, w in levels]
    xc = [torch.randn(B, cin, h, w, generator=g) * 0.5 for h, w in levels]
    torch.manual_seed(1)
    loc_conv, cls_conv = nn.Conv2d(cin, 36, 3, 1, 1), nn.Conv2d(cin, 9 * C, 3, 1, 1)
    tgt = _targets(g, B, 8 if wh[0] > 1000 else 3, C)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, wh if wh[0] != wh[1] else wh[0])

    def run(fused, scale=1.0):
        lc, cc = nn.Conv2d(cin, 36, 3, 1, 1).to(DEV), nn.Conv2d(cin, 9 * C, 3, 1, 1).to(DEV)
        lc.load_state_dict(loc_conv.state_dict()); cc.load_state_dict(cls_conv.state_dict())
        if weights_cl:
            lc, cc = lc.to(memory_format=torch.channels_last), cc.to(memory_format=torch.channels_last)
        xld = [x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for x in xl]
        xcd = [x.to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for x in xc]
        RetinaNetLoss.fuse_head_backward = fused
        _lib.profile_enable("retina_main", True)
        try:
            loc, cls = rh.final_convs(lc, cc, xld, xcd)
            out = crit((loc, cls), tgt.to(DEV))
            (out["loss"] * scale).backward()
        finally:
            RetinaNetLoss.fuse_head_backward = True
            _lib.profile_enable("retina_main", False)
        _lib.profile_collect("retina_main")
        return out, xld + xcd, [lc.weight, lc.bias, cc.weight, cc.bias], (loc, cls)

    of, xf, pf, (locf, clsf) = run(True)
    od, xd, pd, _ = run(False)
    assert int(crit.last_counts[0]) > 0, "no positive anchor: the test would be vacuous"
    for k in of:                                                       # (the 256-channel forward convolutions are not run-to-run identical)
        assert of[k].item() == od[k].item() or (cin >= 256 and abs(of[k].item() - od[k].item()) <= 1e-6 * abs(od[k].item()))
    for a, b in zip(xf, xd):
        assert a.grad.stride() == a.stride()
        _close(a.grad, b.grad)
    for a, b in zip(pf, pd):
        assert a.grad.stride() == a.stride()
        _close(a.grad, b.grad)
    # upstream scale applied on the device; bitwise reproducible
    o3, x3, p3, _ = run(True, 2.5)
    for a, b in zip(x3 + p3, xf + pf):
        _close(a.grad, b.grad * 2.5, 1e-5, 1e-6)
    o4, x4, p4, (loc4, cls4) = run(True)
    if torch.equal(loc4, locf) and torch.equal(cls4, clsf):            # (MIOpen's forward convolution is not always run-to-run identical)
        for a, b in zip(x4 + p4, xf + pf):
            assert torch.equal(a.grad, b.grad)
    else:
        assert cin >= 256, "the small forward convolutions are deterministic"
    # the oracle, end to end on the CPU
    xlc, xcc = [x.clone().requires_grad_() for x in xl], [x.clone().requires_grad_() for x in xc]
    wl, bl, wc, bc = (t.detach().clone().requires_grad_() for t in (loc_conv.weight, loc_conv.bias, cls_conv.weight, cls_conv.bias))
    loc_r, cls_r = _ref_forward(xlc, xcc, wl, bl, wc, bc, C)
    _close(locf, loc_r, 1e-4, 1e-5); _close(clsf, cls_r, 1e-4, 1e-5)
    isz = float(wh[0])
    if wh[0] == wh[1]:
        ref = oracle_retina.retina_loss(loc_r, cls_r, tgt, anchors.cpu(), C, isz)
        ref["loss"].backward()
        assert abs(of["loss"].item() - ref["loss"].item()) <= 1e-4 * abs(ref["loss"].item()) + 1e-6
        for a, b in zip(xf, xlc + xcc):
            _close(a.grad, b.grad)
        for a, b in zip(pf, (wl, bl, wc, bc)):
            _close(a.grad, b.grad)


def test_more_positive_rows_than_staging_slots():
    """With 5 staging slots every later row's contribution is summed in place by the dx launch: same numbers as with slots for all."""
    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    g = torch.Generator().manual_seed(4)
    wh, cin, C, B = (128, 128), 16, 5, 2
    anchors = get_anchor_boxes(torch.Tensor(list(wh)), torch.device(DEV))
    xl = [(torch.randn(B, cin, h, w, generator=g) * 0.5).to(DEV).contiguous(memory_format=torch.channels_last) for h, w in _levels(*wh)]
    xc = [(torch.randn(B, cin, h, w, generator=g) * 0.5).to(DEV).contiguous(memory_format=torch.channels_last) for h, w in _levels(*wh)]
    torch.manual_seed(1)
    lc, cc = nn.Conv2d(cin, 36, 3, 1, 1).to(DEV), nn.Conv2d(cin, 9 * C, 3, 1, 1).to(DEV)
    tgt = _targets(g, B, 3, C).to(DEV)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, 128)
    res = []
    try:
        for rows in (4096, 5):
            _lib.check(_lib.lib().odpl_retina_heads_tune(rows), "tune")
            a = [x.clone().requires_grad_() for x in xl + xc]
            for p in list(lc.parameters()) + list(cc.parameters()):
                p.grad = None
            loc, cls = rh.final_convs(lc, cc, a[:5], a[5:])
            crit((loc, cls), tgt)["loss"].backward()
            assert int(crit.last_counts[0]) > 5
            res.append([t.grad.clone() for t in a] + [lc.weight.grad.clone(), cc.weight.grad.clone()])
    finally:
        _lib.lib().odpl_retina_heads_tune(4096)
    for u, v in zip(*res):
        assert torch.equal(u, v)


def test_no_positive_anchor_at_all():
    """Boxes too small for any anchor to reach IoU 0.5: num_pos = 0, the loss is the reference's (0 / max(1, 0)), and every gradient of the
    structured route is an exact zero tensor (the dense dX rows are still written)."""
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    g = torch.Generator().manual_seed(8)
    wh, cin, C, B = (128, 128), 16, 5, 2
    anchors = get_anchor_boxes(torch.Tensor(list(wh)), torch.device(DEV))
    xs = [(torch.randn(B, cin, h, w, generator=g) * 0.5).to(DEV).contiguous(memory_format=torch.channels_last).requires_grad_() for h, w in _levels(*wh) * 2]
    torch.manual_seed(1)
    lc, cc = nn.Conv2d(cin, 36, 3, 1, 1).to(DEV), nn.Conv2d(cin, 9 * C, 3, 1, 1).to(DEV)
    tgt = torch.tensor([[0., 1., 0.5, 0.5, 0.01, 0.01], [1., 2., 0.3, 0.6, 0.012, 0.008]], device=DEV)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, 128)
    loc, cls = rh.final_convs(lc, cc, xs[:5], xs[5:])
    out = crit((loc, cls), tgt)
    out["loss"].backward()
    assert int(crit.last_counts[0]) == 0 and out["loss"].item() == 0.0
    for t in xs + [lc.weight, lc.bias, cc.weight, cc.bias]:
        assert t.grad is not None and not bool(t.grad.any())


def test_retinanet_model_takes_the_structured_route():
    """The RetinaNet caller model: forward goes through `final_convs`, the criterion writes its gradient on the positive rows only
    (retina_main launches row workgroups only) and the parameters of both final convolutions get the dense route's gradients."""
    import argparse
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    args = argparse.Namespace(model_name="RetinaNet", cls_criterion="focal_loss", coord_criterion="smooth_l1_loss", data_module="synthetic")
    torch.manual_seed(0)
    RetinaNet.img_size = 128
    try:
        model = RetinaNet([str(i) for i in range(4)], args).to(DEV).to(memory_format=torch.channels_last).train()
        g = torch.Generator().manual_seed(2)
        x = torch.rand(2, 3, 128, 128, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
        y = _targets(g, 2, 3, 4).to(DEV)
        res = []
        for fused in (True, False):
            RetinaNetLoss.fuse_head_backward = fused
            try:
                model.zero_grad(set_to_none=True)
                out = model(x)
                assert hasattr(out[0], "_odpl_retina_src") and out[0]._odpl_retina_src is out[1]._odpl_retina_src
                d = model.criterion(out, y)
                d["loss"].backward()
            finally:
                RetinaNetLoss.fuse_head_backward = True
            res.append((d["loss"].item(), {n: p.grad.clone() for n, p in model.named_parameters() if n.startswith(("loc_head.8", "cls_head.8"))}))
        assert abs(res[0][0] - res[1][0]) <= 1e-6 * abs(res[1][0]) and len(res[0][1]) == 4      # (two MIOpen forward passes, not always identical)
        for n in res[0][1]:
            _close(res[0][1][n], res[1][1][n], 1e-3, 1e-4)
        assert all(p.grad is not None for p in model.parameters())
    finally:
        RetinaNet.img_size = 600
