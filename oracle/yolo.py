"""Oracle (test infrastructure): YOLO v2/v3/v4 target assignment + region loss on torch-CPU.

Restates `LightningFunc/accuracy.py:305-380` (build_targets) and
`LightningFunc/losses.py:475-905` (RegionLoss_v4/_v3/_v2 and the multi-scale
wrappers).  Sequential target loop => the reference's "last target in row order
wins" rule on duplicate cells is explicit here instead of an accident of
`index_put_`.  Checked against `tests/golden/yolo_*.npz`.
"""
import numpy as np
import torch
import torch.nn.functional as F

from .boxes import iou_plus1, wh_iou

IGNORE_THRES = 0.5   # losses.py:483
OBJ_SCALE = 1        # losses.py:487
NOOBJ_SCALE = 100    # losses.py:488


def scaled_anchors(anchors, img_dim, grid):
    """`losses.py:499-503`: stride and the division are Python doubles, THEN rounded to f32."""
    stride = img_dim / grid
    return torch.tensor([(aw / stride, ah / stride) for aw, ah in anchors], dtype=torch.float32)


def _wrap(idx, size):
    """Python/torch negative-index rule; an index below -size is the reference's IndexError."""
    if idx < -size:
        raise IndexError(f"index {idx} is out of bounds for dimension with size {size}")
    return idx + size if idx < 0 else idx


def assign_targets(pred_boxes, pred_cls, target, anchors, ignore_thres=IGNORE_THRES):
    """`accuracy.py:305-380`.  Returns the reference's ten dense tensors as a dict.

    pred_boxes [B,A,G,G,4] f32 (cx,cy,w,h in grid units), pred_cls [B,A,G,G,C] f32,
    target [nT,6] f32 (img, cls, x, y, w, h normalised), anchors [A,2] f32 (grid units).
    """
    nB, nA, nG = pred_boxes.shape[0], pred_boxes.shape[1], pred_boxes.shape[2]
    nC = pred_cls.shape[-1]
    nT = target.shape[0]
    out = dict(
        obj_mask=torch.zeros(nB, nA, nG, nG, dtype=torch.uint8),
        noobj_mask=torch.ones(nB, nA, nG, nG, dtype=torch.uint8),
        class_mask=torch.zeros(nB, nA, nG, nG),
        iou_scores=torch.zeros(nB, nA, nG, nG),
        tx=torch.zeros(nB, nA, nG, nG), ty=torch.zeros(nB, nA, nG, nG),
        tw=torch.zeros(nB, nA, nG, nG), th=torch.zeros(nB, nA, nG, nG),
        tcls=torch.zeros(nB, nA, nG, nG, nC),
    )
    tbox = target[:, 2:6] * nG                                   # :327
    gxy, gwh = tbox[:, :2], tbox[:, 2:]
    if nT == 0:
        # the reference's torch.stack(...).max(0) on [A,0] returns empties; nothing is written
        out["tconf"] = out["obj_mask"].float()
        return out
    ious = torch.stack([wh_iou(a, gwh) for a in anchors])        # :331  [A,nT]
    best_n = ious.max(0)[1]                                       # :332  first index on ties
    b = target[:, 0].long().tolist()                              # :334  trunc toward zero
    lab = target[:, 1].long().tolist()
    gi = gxy[:, 0].long().tolist()                                # :337
    gj = gxy[:, 1].long().tolist()
    bn = best_n.tolist()

    # guard 1 (:340-346): all-or-nothing, upper bound only
    g1 = not any(b[t] >= nB or bn[t] >= nA or gj[t] >= nG or gi[t] >= nG for t in range(nT))
    if g1:
        for t in range(nT):
            c = (_wrap(b[t], nB), bn[t], _wrap(gj[t], nG), _wrap(gi[t], nG))
            out["obj_mask"][c] = 1
            out["noobj_mask"][c] = 0
    # ignore loop (:349-358): per-target guard, no check on best_n
    over = (ious > ignore_thres)
    for t in range(nT):
        if b[t] >= nB or gj[t] >= nG or gi[t] >= nG:
            continue
        bb, jj, ii = _wrap(b[t], nB), _wrap(gj[t], nG), _wrap(gi[t], nG)
        for a in range(nA):
            if over[a, t]:
                out["noobj_mask"][bb, a, jj, ii] = 0
    # guard 2 (:361-377): guard 1 plus label < C
    g2 = g1 and not any(l >= nC for l in lab)
    if g2:
        fx = gxy[:, 0] - gxy[:, 0].floor()
        fy = gxy[:, 1] - gxy[:, 1].floor()
        lw = torch.log(gwh[:, 0] / anchors[best_n][:, 0] + 1e-16)
        lh = torch.log(gwh[:, 1] / anchors[best_n][:, 1] + 1e-16)
        cells = [(_wrap(b[t], nB), bn[t], _wrap(gj[t], nG), _wrap(gi[t], nG)) for t in range(nT)]
        idx = tuple(torch.tensor(v) for v in zip(*cells))
        pc = pred_cls[idx]                                         # [nT,C]
        cm = (pc.argmax(-1) == torch.tensor(lab)).float()          # :376 (raw label, as the reference)
        io = iou_plus1(pred_boxes[idx], tbox, x1y1x2y2=False)     # :377
        for t in range(nT):                                        # row order => last writer wins
            c = cells[t]
            out["tx"][c] = fx[t]; out["ty"][c] = fy[t]
            out["tw"][c] = lw[t]; out["th"][c] = lh[t]
            out["tcls"][c + (_wrap(lab[t], nC),)] = 1            # multi-hot on shared cells
            out["class_mask"][c] = cm[t]
            out["iou_scores"][c] = io[t]
    out["tconf"] = out["obj_mask"].float()                        # :379
    return out


def _coord_fn(name):
    return {"smooth_l1_loss": F.smooth_l1_loss, "mse_loss": F.mse_loss}[name]


def _cls_fn(name):
    if name == "bce_loss":
        return F.binary_cross_entropy
    if name == "ce_loss":
        return F.cross_entropy          # nn.CrossEntropyLoss on probability targets
    raise ValueError(name)


def region_loss(head, target, anchors, num_classes, img_dim, num_anchors=None,
                coord="smooth_l1_loss", cls="bce_loss", return_assign=False):
    """One scale: `losses.py:507-575` (v4) == `669-736` (v3) == `830-905` (v2).

    head [B, A*(5+C), G, G] f32 (requires_grad allowed), anchors = list of (w,h) python numbers in
    the units the caller passes (the quirks of v2/v3 live in the wrappers below).
    Returns the 7-tuple (total, lx+ly, size, conf, cls, conf_obj, conf_noobj).
    """
    A = len(anchors) if num_anchors is None else num_anchors
    B, G = head.shape[0], head.shape[2]
    C = num_classes
    p = head.view(B, A, C + 5, G, G).permute(0, 1, 3, 4, 2).contiguous()
    sx, sy = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1])
    w, h = p[..., 2], p[..., 3]
    conf = torch.sigmoid(p[..., 4])
    pcls = torch.sigmoid(p[..., 5:])
    sa = scaled_anchors(anchors, img_dim, G)
    gx = torch.arange(G, dtype=torch.float32).view(1, 1, 1, G)
    gy = torch.arange(G, dtype=torch.float32).view(1, 1, G, 1)
    pb = torch.empty(B, A, G, G, 4)
    pb[..., 0] = sx.detach() + gx
    pb[..., 1] = sy.detach() + gy
    pb[..., 2] = torch.exp(w.detach()) * sa[:, 0].view(1, A, 1, 1)
    pb[..., 3] = torch.exp(h.detach()) * sa[:, 1].view(1, A, 1, 1)
    asg = assign_targets(pb, pcls.detach(), target, sa, IGNORE_THRES)
    om, nm = asg["obj_mask"].bool(), asg["noobj_mask"].bool()
    cfn, kfn = _coord_fn(coord), _cls_fn(cls)
    lx = cfn(sx[om], asg["tx"][om]); ly = cfn(sy[om], asg["ty"][om])
    lw = cfn(w[om], asg["tw"][om]); lh = cfn(h[om], asg["th"][om])
    l_obj = F.binary_cross_entropy(conf[om], asg["tconf"][om])
    l_noobj = F.binary_cross_entropy(conf[nm], asg["tconf"][nm])
    l_conf = OBJ_SCALE * l_obj + NOOBJ_SCALE * l_noobj
    l_cls = kfn(pcls[om], asg["tcls"][om])
    total = lx + ly + lw + lh + l_conf + l_cls
    # "Size" metric (:568-573): raster-ordered predicted wh of obj cells vs target COLUMNS 3:5 (y, w)
    bwbh = pb[..., 2:4][om]
    k = min(len(bwbh), len(target))
    size = cfn(torch.sqrt(torch.abs(bwbh) + 1e-32)[:k], torch.sqrt(torch.abs(target[..., 3:5]) + 1e-32)[:k])
    res = (total, lx + ly, size, l_conf, l_cls, l_obj, l_noobj)
    return (res, asg, pb) if return_assign else res


KEYS = ("loss", "Localization", "Size", "Conf", "Classification", "Conf_obj", "Conf_noobj")


def multiscale_region_loss_v4(heads, target, anchors, anch_masks, num_classes, img_dim=416, **kw):
    """`losses.py:577-634`: anchors picked per scale by mask, every entry averaged over len(heads)."""
    acc = [0.0] * 7
    for i, out in enumerate(heads):
        sub = [anchors[m] for m in anch_masks[i]]
        r = region_loss(out, target, sub, num_classes, img_dim, num_anchors=3, **kw)   # view(.., 3, ..) :519
        acc = [a + v for a, v in zip(acc, r)]
    n = len(heads)
    return dict(zip(KEYS, (a / n for a in acc)))


def multiscale_region_loss_v3(heads, target, anchors_per_scale, num_classes, img_dim=416, **kw):
    """`losses.py:738-795`: anchors arrive already divided by the stride (YOLOV3.py:54-56) and are
    divided by it AGAIN in compute_grid_offsets (`661-665`) -- reproduced, not repaired."""
    acc = [0.0] * 7
    for i, out in enumerate(heads):
        r = region_loss(out, target, anchors_per_scale[i], num_classes, img_dim, num_anchors=3, **kw)
        acc = [a + v for a, v in zip(acc, r)]
    n = len(heads)
    return dict(zip(KEYS, (a / n for a in acc)))


def region_loss_v2(head, target, anchors, num_classes, img_dim=416, **kw):
    """`losses.py:797-905`: single scale, A = len(anchors) (5), dict returned directly (no /n)."""
    r = region_loss(head, target, anchors, num_classes, img_dim, **kw)
    return dict(zip(KEYS, r))


def yolo_statistics(heads, target, anchors, anch_masks, num_classes, img_size, ignore_thres=IGNORE_THRES):
    """`accuracy.py:382-470` (get_yolo_statistics) for the v4 anchor convention (anch_masks given).
    -> {G: [cls_acc, recall50, recall75, precision, conf_obj, conf_noobj, output[B, A*G*G, 5+C]]}."""
    res = {}
    C = num_classes
    for i, x in enumerate(heads):
        sub = [anchors[m] for m in anch_masks[i]]
        A, B, G = len(sub), x.size(0), x.size(2)
        p = x.view(B, A, C + 5, G, G).permute(0, 1, 3, 4, 2).contiguous()
        sx, sy, w, h = torch.sigmoid(p[..., 0]), torch.sigmoid(p[..., 1]), p[..., 2], p[..., 3]
        conf, pcls = torch.sigmoid(p[..., 4]), torch.sigmoid(p[..., 5:])
        stride = img_size / G
        sa = scaled_anchors(sub, img_size, G)
        pb = torch.empty(B, A, G, G, 4)
        pb[..., 0] = sx + torch.arange(G, dtype=torch.float32).view(1, 1, 1, G)
        pb[..., 1] = sy + torch.arange(G, dtype=torch.float32).view(1, 1, G, 1)
        pb[..., 2] = torch.exp(w) * sa[:, 0].view(1, A, 1, 1)
        pb[..., 3] = torch.exp(h) * sa[:, 1].view(1, A, 1, 1)
        asg = assign_targets(pb, pcls, target, sa, ignore_thres)
        om, nm = asg["obj_mask"].bool(), asg["noobj_mask"].bool()
        conf50 = (conf > 0.5).float()
        iou50, iou75 = (asg["iou_scores"] > 0.5).float(), (asg["iou_scores"] > 0.75).float()
        det = conf50 * asg["class_mask"] * asg["tconf"]
        out = torch.cat((pb.view(B, -1, 4) * stride, conf.view(B, -1, 1), pcls.view(B, -1, C)), -1)
        res[G] = [(100 * asg["class_mask"][om].mean()).numpy(), (torch.sum(iou50 * det) / (om.sum() + 1e-16)).numpy(),
                  (torch.sum(iou75 * det) / (om.sum() + 1e-16)).numpy(), (torch.sum(iou50 * det) / (conf50.sum() + 1e-16)).numpy(),
                  conf[om].mean().numpy(), conf[nm].mean().numpy(), out]
    return res
