"""Full-size checks through properties that do not need the (slow) oracle at that size, plus oracle parity where it is affordable:
decode at BASELINE config 2's size, IoU algebra on a million boxes, greedy NMS invariants at config 4's size (200 700 anchors),
the SSD loss at its native size (8 732 priors) against the oracle, evaluation statistics on a 16-image batch against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
A18 = [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401]
MASKS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def test_decode_at_config2_size_vs_oracle():
    """yolo_forward_dynamic x 3 + get_region_boxes (utils/YoloV4Utils.py:18-176) on [16, 255, G, G], G = 52 / 26 / 13: 10 647 candidates per
    image x 16 images against the oracle (hardware exp2 / rcp inside: 1e-5 relative, as the small golden case)."""
    from objectdetectionpl_amd.LightningFunc.utils import YoloV4Utils as U
    from oracle import decode as oracle_decode
    g = torch.Generator().manual_seed(5)
    heads = [torch.randn(16, 255, G, G, generator=g) * 1.5 for G in (52, 26, 13)]
    rb, rc = oracle_decode.decode_heads(heads, 80, A18, MASKS, [8, 16, 32])
    b, c = U.decode_heads([h.to(DEV) for h in heads], 80, A18, MASKS, [8, 16, 32])
    assert b.shape == rb.shape == (16, 10647, 1, 4) and c.shape == rc.shape == (16, 10647, 80)
    assert torch.allclose(b.cpu(), rb, rtol=1e-5, atol=1e-6) and torch.allclose(c.cpu(), rc, rtol=1e-5, atol=1e-7)


def test_iou_algebra_on_a_million_boxes():
    """bbox_iou (accuracy.py:6-34, the "+1" pixel form) and bbox_iou_v5 (accuracy.py:71-114): symmetric in their arguments bit for bit,
    1 on identical boxes, inside [0, 1]; GIoU <= IoU, DIoU <= IoU, CIoU <= DIoU; the IoU matrix row of a box equals the pair IoU."""
    from objectdetectionpl_amd.LightningFunc import accuracy as A
    g = torch.Generator().manual_seed(9)
    n = 1 << 20
    def boxes():
        xy = torch.rand(n, 2, generator=g) * 400
        wh = torch.rand(n, 2, generator=g) * 120 + 1
        return torch.cat([xy, xy + wh], 1).to(DEV)
    a, b = boxes(), boxes()
    i_ab, i_ba = A.bbox_iou(a, b), A.bbox_iou(b, a)
    assert torch.equal(i_ab, i_ba) and float(i_ab.min()) >= 0 and float(i_ab.max()) <= 1
    assert torch.equal(A.bbox_iou(a, a), torch.ones(n, device=DEV))
    v = {m: A.bbox_iou_v5(a.t(), b.t(), **({m: True} if m != "IoU" else {})) for m in ("IoU", "GIoU", "DIoU", "CIoU")}   # both [4, n] (accuracy.py:73)
    assert torch.equal(v["IoU"], A.bbox_iou_v5(b.t(), a.t()))
    eps = 1e-6
    assert bool((v["GIoU"] <= v["IoU"] + eps).all()) and bool((v["DIoU"] <= v["IoU"] + eps).all()) and bool((v["CIoU"] <= v["DIoU"] + eps).all())
    assert float(v["GIoU"].min()) >= -1 - eps and float(v["IoU"].min()) >= 0
    m = A.box_iou_matrix(a[:64], b[:4096])
    assert m.shape == (64, 4096)
    pair = A.box_iou_matrix(a[:64], b[:64]).diagonal()
    assert torch.equal(m[:, :64].diagonal(), pair)


def test_greedy_nms_at_config4_size_vs_oracle_and_invariants():
    """model/RetinaNet.py:117-178 on 4 x 200 700 anchors, C = 80, against the oracle's sequential loop: the same rows in the same order.
    (The reference indexes the UNFILTERED box / label arrays with positions in the FILTERED candidate list -- kept, so "no kept pair
    overlaps" is NOT a property of its output.)  Invariants that do hold: at most topk rows, scores descending and above the threshold."""
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    from oracle import retina as oracle_retina
    g = torch.Generator().manual_seed(21)
    anchors = get_anchor_boxes(torch.Tensor([1333, 800]), torch.device(DEV))
    An = anchors.shape[0]
    assert An == 200700
    B, C = 4, 80
    loc = torch.randn(B, An, 4, generator=g) * 0.2
    cls = torch.randn(B, An, C, generator=g) * 1.2 - 3.5                    # a few thousand candidates per image above sigmoid^-1(0.45)
    out = greedy_non_max_suppression((loc.to(DEV), cls.to(DEV)), anchors, topk=100, nms_thresh=0.5, class_thresh=0.45)
    ref = oracle_retina.greedy_nms(loc, cls, anchors.cpu(), topk=100, nms_thresh=0.5, class_thresh=0.45)
    assert len(out) == len(ref) == B
    for det, r in zip(out, ref):
        k = det.shape[0]
        assert 0 < k <= 100 and det.shape == r.shape
        sc = det[:, 5]
        assert bool((sc[:-1] >= sc[1:]).all()) and float(sc.min()) > 0.45
        assert torch.equal(det[:, 6].cpu(), r[:, 6])                                        # labels: index work
        assert torch.allclose(det.cpu(), r, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("cls_name", ["ce_loss", "focal_loss"])
def test_ssd_loss_at_native_size_vs_oracle(cls_name):
    """SSDLoss (losses.py:159-315) at its native size: 8 732 priors, 21 classes, B = 16, ~6 boxes per image -- scalars, the support of
    both gradients (matched priors + 3 : 1 hard negatives: index work) and their values against the oracle."""
    import torch.nn as nn
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils.SSDUtils import get_dboxes
    from oracle import ssd as oracle_ssd
    g = torch.Generator().manual_seed(33)
    B, C, P = 16, 21, 8732
    priors = get_dboxes().cpu()
    assert priors.shape == (P, 4)
    off = torch.randn(B, P, 4, generator=g) * 0.5
    cls = torch.randn(B, P, C, generator=g)
    n = B * 6
    wh = torch.rand(n, 2, generator=g) * 0.4 + 0.08
    tgt = torch.cat([torch.arange(B).repeat_interleave(6).float()[:, None], torch.randint(1, C, (n, 1), generator=g).float(),
                     torch.rand(n, 2, generator=g) * (1 - wh), wh], 1)
    oc, cc = off.clone().requires_grad_(), cls.clone().requires_grad_()
    ref = oracle_ssd.ssd_loss(oc, cc, tgt, priors, cls=cls_name)
    ref["loss"].backward()
    crit = SSDLoss(priors, focal_loss if cls_name == "focal_loss" else nn.CrossEntropyLoss, "smooth_l1_loss", C, 300)
    od, cd = off.to(DEV).requires_grad_(), cls.to(DEV).requires_grad_()
    out = crit((od, cd), tgt.to(DEV))
    out["loss"].backward()
    for k in ("loss", "Localization", "Classification"):
        a, b = out[k].item(), ref[k].item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (k, a, b)
    for got, want in ((od.grad.cpu(), oc.grad), (cd.grad.cpu(), cc.grad)):
        assert torch.equal(got != 0, want != 0)
        assert torch.allclose(got, want, rtol=1e-4, atol=1e-6 * float(want.abs().max()) + 1e-12)


def test_batch_statistics_and_ap_on_a_16_image_batch_vs_oracle():
    """get_batch_statistics + ap_per_class (accuracy.py:116-154, 207-287) on 16 images x 300 detections x 12 targets, 20 classes."""
    from objectdetectionpl_amd.LightningFunc.accuracy import ap_per_class, get_batch_statistics
    from oracle import metrics as oracle_metrics
    g = torch.Generator().manual_seed(41)
    B, K, C = 16, 300, 20
    outs, tg = [], []
    for b in range(B):
        nt = 12
        txy = torch.rand(nt, 2, generator=g) * 300 + 20
        twh = torch.rand(nt, 2, generator=g) * 80 + 20
        tcls = torch.randint(0, C, (nt, 1), generator=g).float()
        tg.append(torch.cat([torch.full((nt, 1), float(b)), tcls, txy, txy + twh], 1))          # (img, cls, x1, y1, x2, y2) as step.py hands it over
        src = torch.randint(0, nt, (K,), generator=g)
        box = torch.cat([txy[src], (txy + twh)[src]], 1) + torch.randn(K, 4, generator=g) * 12
        score = torch.rand(K, generator=g)
        lab = torch.where(torch.rand(K, generator=g) < 0.7, tcls[src, 0], torch.randint(0, C, (K,), generator=g).float())
        outs.append(torch.cat([box, score[:, None], score[:, None], lab[:, None]], 1))       # distinct confidences: no sort ties
    targets = torch.cat(tg, 0)
    ref = oracle_metrics.batch_statistics([o.clone() for o in outs], targets.clone(), 0.5)
    got = get_batch_statistics([o.to(DEV) for o in outs], targets.to(DEV), 0.5)
    assert len(ref) == len(got) == B
    for (tp, sc, lb), (rtp, rsc, rlb) in zip(got, ref):
        assert np.array_equal(tp, np.asarray(rtp)) and np.array_equal(sc, np.asarray(rsc)) and np.array_equal(lb, np.asarray(rlb))
    tp = np.concatenate([s[0] for s in got]); conf = np.concatenate([s[1] for s in got]); pc = np.concatenate([s[2] for s in got])
    tc = targets[:, 1].numpy()
    r = oracle_metrics.ap_per_class(tp, conf, pc, tc)
    o = ap_per_class(tp, conf, pc, tc)
    for a, b in zip(o, r):
        assert np.allclose(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64), rtol=1e-12, atol=1e-15)
