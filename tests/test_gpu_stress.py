"""GPU parity at awkward sizes: many targets (beyond every LDS staging capacity), one class, many classes,
odd grids, batch 1 -- each against the CPU oracle."""
import numpy as np
import pytest
import torch

from oracle import retina as oracle_retina
from oracle import yolo as oracle_yolo
from oracle import yolov5 as oracle_v5

pytestmark = pytest.mark.gpu
A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
M3 = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
V5A = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def _targets(g, B, n, C, lo=0.02, hi=0.5):
    wh = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    return torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), torch.randint(0, C, (n, 1), generator=g).float(),
                      torch.rand(n, 2, generator=g) * (1 - wh), wh], 1)


def _close(a, b, rtol=1e-4):
    a = a.detach().cpu().double(); b = b.detach().cpu().double()
    scale = float(b.abs().max()) if b.numel() else 0.0
    return torch.allclose(a, b, rtol=rtol, atol=1e-6 * scale + 1e-12, equal_nan=True)


@pytest.mark.parametrize("B,C,grids,nT", [(1, 1, (13,), 3), (3, 200, (19, 10, 5), 40), (2, 4, (52, 26, 13), 4500), (5, 33, (7, 6, 3), 0)])
def test_yolo_loss_awkward_sizes(B, C, grids, nT):
    from objectdetectionpl_amd.LightningFunc.losses import MultiScaleRegionLoss_v4
    g = torch.Generator().manual_seed(B * 1000 + C)
    masks = M3[:len(grids)]
    heads = [torch.randn(B, 3 * (5 + C), G, G, generator=g) for G in grids]
    tgt = _targets(g, B, nT, C)
    ho = [h.clone().requires_grad_() for h in heads]
    ref = oracle_yolo.multiscale_region_loss_v4(ho, tgt, A9, masks, C, 416)
    if nT:
        ref["loss"].backward()
    d = torch.device("cuda:0")
    hd = [h.to(d).requires_grad_() for h in heads]
    crit = MultiScaleRegionLoss_v4(A9, masks, "bce_loss", "smooth_l1_loss", "bce_loss", C, 416)
    out = crit(hd, tgt.to(d))
    out["loss"].backward()
    for k in oracle_yolo.KEYS:
        a, b = out[k].item(), ref[k].item()
        assert (np.isnan(a) and np.isnan(b)) or abs(a - b) <= 1e-4 * abs(b) + 1e-6, (k, a, b)
    if nT:
        for h, o in zip(hd, ho):
            assert _close(h.grad, o.grad)


def test_retina_many_boxes_per_image():
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    d = torch.device("cuda:0")
    anchors = get_anchor_boxes([160, 96], d)
    An, C, B = anchors.shape[0], 7, 3
    g = torch.Generator().manual_seed(77)
    loc = torch.randn(B, An, 4, generator=g) * 0.3
    cls = torch.randn(B, An, C, generator=g)
    n = 900                                        # ~300 boxes per image: several 256-row LDS tiles
    wh = torch.rand(n, 2, generator=g) * 0.5 + 0.05
    tgt = torch.cat([(torch.arange(n) % B).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                     torch.rand(n, 2, generator=g) * 0.8 + 0.1, wh], 1)
    lo = loc.clone().requires_grad_(); co = cls.clone().requires_grad_()
    ref = oracle_retina.retina_loss(lo, co, tgt, anchors.cpu(), C, 128)
    ref["loss"].backward()
    ld = loc.to(d).requires_grad_(); cd = cls.to(d).requires_grad_()
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, 128)
    out = crit((ld, cd), tgt.to(d)); out["loss"].backward()
    for k in ref:
        assert abs(out[k].item() - ref[k].item()) <= 1e-4 * abs(ref[k].item()) + 1e-6, k
    assert _close(cd.grad, co.grad) and _close(ld.grad, lo.grad)
    _, cls_t = oracle_retina.encode_targets(anchors.cpu(), tgt, B, 128)
    assert int(crit.last_counts[0]) == int((cls_t > 0).sum())


@pytest.mark.parametrize("B,C,grids,nT", [(1, 1, (8, 4, 2), 5), (2, 150, (12, 6, 3), 30), (3, 6, (20, 10, 5), 700)])
def test_v5_loss_awkward_sizes(B, C, grids, nT):
    from objectdetectionpl_amd.LightningFunc.losses_dense import MultiScaleRegionLoss_v5
    g = torch.Generator().manual_seed(B * 100 + C)
    heads = [torch.randn(B, 3, G, G, 5 + C, generator=g) for G in grids]
    tgt = _targets(g, B, nT, C, 0.02, 0.6)
    tgt[:, 2:4] = tgt[:, 2:4] * 0.9 + 0.05
    ho = [h.clone().requires_grad_() for h in heads]
    ref = oracle_v5.v5_loss(ho, tgt, V5A, C)
    ref["loss"].backward()
    d = torch.device("cuda:0")
    hd = [h.to(d).requires_grad_() for h in heads]
    crit = MultiScaleRegionLoss_v5(V5A, None, None, None, None, C, 160)
    out = crit(hd, tgt.to(d)); out["loss"].backward()
    for k in ref:
        assert abs(out[k].item() - ref[k].item()) <= 1e-4 * abs(ref[k].item()) + 1e-6, (k, out[k].item(), ref[k].item())
    for h, o in zip(hd, ho):
        assert _close(h.grad, o.grad)


def test_yolo_nms_single_scale_five_anchors_v2_style():
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    from oracle import nms as oracle_nms
    g = torch.Generator().manual_seed(5)
    B, A, C, G = 2, 5, 20, 9
    h = torch.randn(B, A, 5 + C, G, G, generator=g)
    h[:, :, 0:2] = torch.rand(B, A, 2, 1, 1, generator=g) * 200 + 40 + torch.randn(B, A, 2, G, G, generator=g) * 20
    h[:, :, 2:4] = torch.rand(B, A, 2, G, G, generator=g) * 60 + 20
    h[:, :, 5:] = torch.rand(B, A, C, G, G, generator=g)
    h = h.view(B, A * (5 + C), G, G).contiguous()
    ref = oracle_nms.yolo_weighted_nms([h], num_anchors=5)
    out = yolo_non_max_suppression([h.cuda()], num_anchors=5)
    for r, o in zip(ref, out):
        assert o.shape == r.shape and torch.equal(o.cpu()[:, 4:], r[:, 4:]) and torch.allclose(o.cpu()[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)
