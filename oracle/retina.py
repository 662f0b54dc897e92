"""Oracle (test infrastructure): RetinaNet anchors, target encoding, softmax-focal + SmoothL1 loss and
the greedy top-k NMS, restated on torch-CPU from `LightningFunc/utils/RetinaUtils.py:6-80`,
`LightningFunc/losses.py:317-472` and `model/RetinaNet.py:117-178`.  Pinned by tests/golden/retina_*.npz.
"""
import math

import torch
import torch.nn.functional as F

from .boxes import iou_matrix_plus1


def anchor_wh_table():
    """RetinaUtils.py:62-80: python-double maths, then one rounding to f32.  [5,9,2]."""
    out = []
    for s in (32 * 32., 64 * 64., 128 * 128., 256 * 256., 512 * 512.):
        for ar in (1 / 2., 1 / 1., 2 / 1.):
            h = math.sqrt(s / ar)
            w = ar * h
            for sr in (1., pow(2, 1 / 3.), pow(2, 2 / 3.)):
                out.append([w * sr, h * sr])
    return torch.tensor(out, dtype=torch.float32).view(5, 9, 2)


def anchor_boxes(input_size):
    """RetinaUtils.py:6-31.  input_size: f32 tensor (w, h).  -> [sum fm_h*fm_w*9, 4] (cx,cy,w,h)."""
    wh_tab = anchor_wh_table()
    boxes = []
    for i in range(5):
        fm = (input_size / pow(2., i + 3)).ceil()
        grid = input_size / fm
        fw, fh = int(fm[0]), int(fm[1])
        xs = torch.arange(0, fw, dtype=torch.float).repeat(fh)
        ys = torch.arange(0, fh, dtype=torch.float).repeat_interleave(fw)
        xy = (torch.stack([xs, ys], 1) + 0.5) * grid                      # [fh*fw, 2]
        xy = xy.view(fh, fw, 1, 2).expand(fh, fw, 9, 2)
        wh = wh_tab[i].view(1, 1, 9, 2).expand(fh, fw, 9, 2)
        boxes.append(torch.cat([xy, wh], 3).reshape(-1, 4))
    return torch.cat(boxes, 0)


def encode_targets(anchors, targets, batch_size, img_size):
    """losses.py:423-446: per image max-IoU assignment.  -> loc_targets [B,An,4], cls_targets [B,An] int64."""
    locs, clss = [], []
    for bid in range(batch_size):
        sel = targets[:, 0] == bid
        # losses.py:425 multiplies by ONE scalar; a (width, height) pair is the build's documented extension for non-square
        # inputs (x / w by width, y / h by height) and has no counterpart in the reference
        scale = torch.tensor([img_size[0], img_size[1], img_size[0], img_size[1]], dtype=torch.float32) if isinstance(img_size, (tuple, list)) else img_size
        boxes = targets[sel][:, 2:] * scale
        labels = targets[sel][:, 1].long()
        ious = iou_matrix_plus1(anchors, boxes, order="xywh")
        max_ious, max_ids = ious.max(1)
        bx = boxes[max_ids]
        loc = torch.cat([(bx[:, :2] - anchors[:, :2]) / anchors[:, 2:], torch.log(bx[:, 2:] / anchors[:, 2:])], 1)
        cls_t = 1 + labels[max_ids]
        cls_t[max_ious < 0.5] = 0
        cls_t[(max_ious > 0.4) & (max_ious < 0.5)] = -1
        locs.append(loc)
        clss.append(cls_t)
    return torch.stack(locs), torch.stack(clss)


def softmax_focal(x, y, num_classes, reduction="sum"):
    """losses.py:317-349 (class focal_loss): SOFTMAX focal, alpha .25, gamma 2, background row = zeros."""
    t = torch.eye(num_classes + 1)[y.long()][:, 1:]
    p = F.softmax(x, dim=1).clamp(1e-7, 1. - 1e-7)
    l = 0.25 * (-1 * t * torch.log(p)) * (1 - p) ** 2
    return l.sum() if reduction == "sum" else l


def retina_loss(loc_preds, cls_preds, targets, anchors, num_classes, img_size, coord="smooth_l1_loss"):
    """losses.py:405-472.  Returns dict(loss, Localization, Classification)."""
    B = loc_preds.shape[0]
    loc_t, cls_t = encode_targets(anchors, targets, B, img_size)
    pos = cls_t > 0
    num_pos = pos.long().sum()
    m = pos.unsqueeze(2).expand_as(loc_preds)
    fn = {"smooth_l1_loss": F.smooth_l1_loss, "mse_loss": F.mse_loss}[coord]
    loc_loss = fn(loc_preds[m].view(-1, 4), loc_t[m].view(-1, 4), reduction="sum")
    pn = cls_t > -1
    mc = pn.unsqueeze(2).expand_as(cls_preds)
    cls_loss = softmax_focal(cls_preds[mc].view(-1, num_classes), cls_t[pn], num_classes, "sum")
    n = max(1.0, num_pos.item())
    return {"loss": (loc_loss + cls_loss) / n, "Localization": loc_loss / n, "Classification": cls_loss / n}


def decode_boxes(loc, anchors):
    """RetinaNet.py:121-126: -> [An,4] x1,y1,x2,y2."""
    xy = loc[:, :2] * anchors[:, 2:] + anchors[:, :2]
    wh = loc[:, 2:].exp() * anchors[:, 2:]
    return torch.cat([xy - wh / 2, xy + wh / 2], 1)


def greedy_nms(loc_preds, cls_preds, anchors, topk=100, nms_thresh=0.5, class_thresh=0.45):
    """RetinaNet.py:117-178 == SSD.py:249-310, quirks kept:
    * `keep` indexes the FILTERED candidate list but boxes/labels are taken from the UNFILTERED arrays;
    * the loop stops without keeping the last box when exactly one remains;
    * exactly one candidate above the threshold is an IndexError in the reference (0-dim squeeze)."""
    out = []
    for bid in range(loc_preds.size(0)):
        boxes = decode_boxes(loc_preds[bid], anchors)
        score, labels = cls_preds[bid].sigmoid().max(1)
        ids = (score > class_thresh).nonzero().flatten()
        if ids.numel() == 1:
            raise IndexError("too many indices for tensor of dimension 1")
        bb, sc = boxes[ids], score[ids]
        x1, y1, x2, y2 = bb[:, 0], bb[:, 1], bb[:, 2], bb[:, 3]
        areas = (x2 - x1 + 1) * (y2 - y1 + 1)
        # (stable: among equal scores the first in index order -- the reference's plain sort leaves ties unspecified, RetinaNet.py:140;
        # its fixtures hold no ties, the kernels define them like this)
        order = sc.sort(dim=0, descending=True, stable=True)[1][:topk].tolist()
        keep = []
        while len(order) > 1:
            i = order[0]
            keep.append(i)
            rest = torch.tensor(order[1:])
            w = (x2[rest].clamp(max=x2[i]) - x1[rest].clamp(min=x1[i]) + 1).clamp(min=0)
            h = (y2[rest].clamp(max=y2[i]) - y1[rest].clamp(min=y1[i]) + 1).clamp(min=0)
            inter = w * h
            ovr = inter / (areas[i] + areas[rest] - inter)
            order = rest[ovr <= nms_thresh].tolist()
        keep = torch.tensor(keep, dtype=torch.long)
        out.append(torch.cat([boxes[keep], torch.zeros(len(keep), 1), sc[keep].unsqueeze(1),
                              labels[keep].unsqueeze(1).float()], 1))
    return out
