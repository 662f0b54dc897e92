"""Oracle (test infrastructure): the IoU family of the reference, restated on torch-CPU.

Each function keeps the reference's fp32 operation ORDER (results are compared
bit-for-bit against `tests/golden/iou.npz`, captured from the reference itself).
"""
import math

import torch


def corners_from_centre(box):
    """(cx,cy,w,h) -> x1,y1,x2,y2 as four tensors; `accuracy.py:43-48` / `289-295`."""
    hw, hh = box[..., 2] / 2, box[..., 3] / 2
    return box[..., 0] - hw, box[..., 1] - hh, box[..., 0] + hw, box[..., 1] + hh


def xywh2xyxy(x):
    """`accuracy.py:289-295`."""
    x1, y1, x2, y2 = corners_from_centre(x)
    return torch.stack((x1, y1, x2, y2), -1)


def wh_iou(anchor_wh, gwh):
    """Anchor-shape IoU, `accuracy.py:297-303` (bbox_wh_iou).

    anchor_wh: [2] f32, gwh: [nT,2] f32 -> [nT].  The 1e-16 joins the anchor
    area first, exactly as the reference associates it.
    """
    w1, h1 = anchor_wh[0], anchor_wh[1]
    w2, h2 = gwh[:, 0], gwh[:, 1]
    inter = torch.min(w1, w2) * torch.min(h1, h2)
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter
    return inter / union


def iou_plus1(box1, box2, x1y1x2y2=True):
    """Pixel-convention ("+1") IoU, `accuracy.py:39-69` (bbox_iou).

    The reference accepts GIoU/DIoU/CIoU keywords here and ignores them.
    """
    if x1y1x2y2:
        a = (box1[:, 0], box1[:, 1], box1[:, 2], box1[:, 3])
        b = (box2[:, 0], box2[:, 1], box2[:, 2], box2[:, 3])
    else:
        a, b = corners_from_centre(box1), corners_from_centre(box2)
    ix1, iy1 = torch.max(a[0], b[0]), torch.max(a[1], b[1])
    ix2, iy2 = torch.min(a[2], b[2]), torch.min(a[3], b[3])
    inter = torch.clamp(ix2 - ix1 + 1, min=0) * torch.clamp(iy2 - iy1 + 1, min=0)
    area_a = (a[2] - a[0] + 1) * (a[3] - a[1] + 1)
    area_b = (b[2] - b[0] + 1) * (b[3] - b[1] + 1)
    return inter / (area_a + area_b - inter + 1e-16)


def iou_v5(box1, box2, x1y1x2y2=True, mode="iou"):
    """Transposed-layout IoU / GIoU / DIoU / CIoU, `accuracy.py:71-114` (bbox_iou_v5).

    box1, box2: [4,N].  mode in {"iou","giou","diou","ciou"} (the reference's
    three booleans; GIoU takes precedence there, then DIoU, then CIoU).
    """
    if x1y1x2y2:
        a = (box1[0], box1[1], box1[2], box1[3])
        b = (box2[0], box2[1], box2[2], box2[3])
    else:
        a = (box1[0] - box1[2] / 2, box1[1] - box1[3] / 2, box1[0] + box1[2] / 2, box1[1] + box1[3] / 2)
        b = (box2[0] - box2[2] / 2, box2[1] - box2[3] / 2, box2[0] + box2[2] / 2, box2[1] + box2[3] / 2)
    inter = (torch.min(a[2], b[2]) - torch.max(a[0], b[0])).clamp(0) * \
            (torch.min(a[3], b[3]) - torch.max(a[1], b[1])).clamp(0)
    w1, h1 = a[2] - a[0], a[3] - a[1]
    w2, h2 = b[2] - b[0], b[3] - b[1]
    union = (w1 * h1 + 1e-16) + w2 * h2 - inter
    iou = inter / union
    if mode == "iou":
        return iou
    cw = torch.max(a[2], b[2]) - torch.min(a[0], b[0])
    ch = torch.max(a[3], b[3]) - torch.min(a[1], b[1])
    if mode == "giou":
        c_area = cw * ch + 1e-16
        return iou - (c_area - union) / c_area
    c2 = cw ** 2 + ch ** 2 + 1e-16
    rho2 = ((b[0] + b[2]) - (a[0] + a[2])) ** 2 / 4 + ((b[1] + b[3]) - (a[1] + a[3])) ** 2 / 4
    if mode == "diou":
        return iou - rho2 / c2
    if mode == "ciou":
        v = (4 / math.pi ** 2) * torch.pow(torch.atan(w2 / h2) - torch.atan(w1 / h1), 2)
        with torch.no_grad():
            alpha = v / (1 - iou + v)
        return iou - (rho2 / c2 + v * alpha)
    raise ValueError(mode)


def iou_corner(t1, t2):
    """Corner IoU without "+1" and without epsilon, `accuracy.py:6-37` (iou), used by SSD matching.

    t1, t2: same shape [..., 4] (x1,y1,x2,y2).
    """
    lo = torch.max(t1[..., :2], t2[..., :2])
    hi = torch.min(t1[..., 2:], t2[..., 2:])
    d = torch.clamp(hi - lo, min=0.0)
    inter = d[..., 0] * d[..., 1]
    d1 = torch.clamp(t1[..., 2:] - t1[..., :2], min=0.0)
    d2 = torch.clamp(t2[..., 2:] - t2[..., :2], min=0.0)
    return inter / (d1[..., 0] * d1[..., 1] + d2[..., 0] * d2[..., 1] - inter)


def iou_matrix_plus1(box1, box2, order="xyxy"):
    """[N,4] x [M,4] -> [N,M] "+1" IoU of RetinaNetLoss, `losses.py:375-403` (+ change_box_order 360-373)."""
    if order == "xywh":
        box1 = torch.cat([box1[:, :2] - box1[:, 2:] / 2, box1[:, :2] + box1[:, 2:] / 2], 1)
        box2 = torch.cat([box2[:, :2] - box2[:, 2:] / 2, box2[:, :2] + box2[:, 2:] / 2], 1)
    lt = torch.max(box1[:, None, :2], box2[:, :2])
    rb = torch.min(box1[:, None, 2:], box2[:, 2:])
    wh = (rb - lt + 1).clamp(min=0)
    inter = wh[:, :, 0] * wh[:, :, 1]
    area1 = (box1[:, 2] - box1[:, 0] + 1) * (box1[:, 3] - box1[:, 1] + 1)
    area2 = (box2[:, 2] - box2[:, 0] + 1) * (box2[:, 3] - box2[:, 1] + 1)
    return inter / (area1[:, None] + area2 - inter)
