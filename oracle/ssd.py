"""Oracle (test infrastructure): SSD matching + hard-negative loss, restated on torch-CPU from
`LightningFunc/losses.py:159-315` (SSDLoss) and `accuracy.py:6-37` (iou); priors from `utils/SSDUtils.py:5-27`.
Pinned by tests/golden/ssd.npz."""
import itertools
import math

import torch
import torch.nn.functional as F

from .boxes import iou_corner
from .retina import softmax_focal


def dboxes(smin=0.07, smax=0.9, ars=(1, 2, 0.5, 3, 1 / 3.0), fks=(38, 19, 10, 5, 3, 1), num_boxes=(3, 5, 5, 5, 3, 3)):
    """SSDUtils.py:5-27."""
    m = len(fks)
    sks = [round(smin + (((smax - smin) / (m - 1)) * (k - 1)), 2) for k in range(1, m + 1)]
    out = []
    for k, fk in enumerate(fks):
        for i, j in itertools.product(range(fk), range(fk)):
            cx, cy = (i + 0.5) / fk, (j + 0.5) / fk
            s = math.sqrt(sks[k] * sks[min(k + 1, m - 1)])
            out.append([cx, cy, s, s])
            for ar in ars[:num_boxes[k]]:
                out.append([cx, cy, sks[k] * math.sqrt(ar), sks[k] / math.sqrt(ar)])
    return torch.clamp(torch.tensor(out).float(), max=1.0)


def to_points(c):
    """losses.py:172-185: clamp the corners into [0, 1]."""
    return torch.cat([torch.clamp(c[:, :2] - c[:, 2:] / 2.0, min=0.0), torch.clamp(c[:, :2] + c[:, 2:] / 2.0, max=1.0)], 1)


def match(priors, boxes, thresh=0.5):
    """losses.py:199-218.  -> (annotation index per prior, matched flag per prior)."""
    M = boxes.size(0)
    pp, bp = to_points(priors), to_points(boxes)
    ious = iou_corner(pp.unsqueeze(0).expand(M, -1, -1), bp.unsqueeze(1).expand(M, pp.size(0), 4))
    best_prior = ious.max(1)[1]
    ious_max, ann = ious.max(0)
    matched = ious_max >= thresh
    for m in range(M):                       # forced matches in annotation order: the last one wins a shared prior
        matched[best_prior[m]] = True
        ann[best_prior[m]] = m
    return ann, matched


def image_loss(priors, classes, boxes, pred_cls, pred_off, cls="ce_loss", coord="smooth_l1_loss", neg_ratio=3):
    """losses.py:238-275 (annotations non-empty)."""
    ann, matched = match(priors, boxes)
    pos = matched.nonzero().squeeze(1)
    neg = (matched == 0).nonzero().squeeze(1)
    N = pos.size(0)
    mb = boxes[ann]
    true_off = torch.cat([(mb[:, :2] - priors[:, :2]) / (priors[:, 2:] * 0.1), torch.log(mb[:, 2:] / priors[:, 2:]) / 0.2], 1)
    cfn = {"smooth_l1_loss": F.smooth_l1_loss, "mse_loss": F.mse_loss}[coord]
    reg = cfn(pred_off[pos], true_off[pos], reduction="none")
    true_cls = torch.zeros(pred_cls.size(0), dtype=torch.long)
    true_cls[pos] = classes.long()[ann[pos]]
    if cls == "ce_loss":
        tot = F.cross_entropy(pred_cls, true_cls, reduction="none")
        hard = tot[neg].sort(descending=True)[1][:N * neg_ratio]
        cl = (tot[pos].sum() + tot[neg][hard].sum()) / N
    else:
        # focal_loss(reduction='none') is a [P, C] matrix; the reference then sorts ROWS and gathers whole rows
        # (losses.py:264-272).  Non-matched priors are background => their rows are exactly 0, so the
        # "hard negative" term is 0 and only the positives' entries remain.
        tot = softmax_focal(pred_cls, true_cls, pred_cls.size(1), "none")
        cl = tot[pos].sum() / N
    return cl, reg.sum() / N


def ssd_loss(pred_off, pred_cls, targets, priors, cls="ce_loss", coord="smooth_l1_loss"):
    """losses.py:277-315, quirk kept: image j is scored against the FIRST lens[j] target rows of the whole batch."""
    B = pred_cls.size(0)
    x = targets[:, 0]
    lens = [(x == u).sum().item() for u in x.unique(sorted=True)]
    cl_sum, loc_sum, nb = 0.0, 0.0, B
    for j in range(B):
        if j >= len(lens):
            nb -= 1
            continue
        a = targets[:lens[j]]
        c, l = image_loss(priors, a[:, 1], a[:, 2:], pred_cls[j], pred_off[j], cls, coord)
        cl_sum = cl_sum + c
        loc_sum = loc_sum + l
    loc, cl = loc_sum / nb, cl_sum / nb
    return {"loss": loc + cl, "Localization": loc, "Classification": cl}
