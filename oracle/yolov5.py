"""Oracle (test infrastructure): YOLOv5 target assignment and loss, restated on torch-CPU from
`LightningFunc/accuracy.py:472-521` (build_targets_v5), `LightningFunc/losses.py:37-156`
(FocalLoss, MultiScaleRegionLoss_v5) and `accuracy.py:71-114` (bbox_iou_v5).  Pinned by tests/golden/yolov5.npz."""
import torch
import torch.nn.functional as F

from .boxes import iou_v5

STRIDES = (8.0, 16.0, 32.0)          # losses.py:73
ANCHOR_T = 4.0                        # accuracy.py:478
GAIN_BOX, GAIN_OBJ, GAIN_CLS = 0.05, 1.0, 0.58    # losses.py:137-139


def grid_anchors(anchors):
    """losses.py:95-96: pixel anchors [[w0,h0,w1,h1,w2,h2], ...] -> [nl, na, 2] in grid units (f32 division)."""
    a = torch.tensor(anchors).float().view(len(anchors), -1, 2)
    return a / torch.tensor(STRIDES[:len(anchors)]).view(-1, 1, 1)


def assign_layer(G, targets, anc):
    """accuracy.py:481-519 for one layer ('rect4').  Row order = [base rows (anchor-major, target order),
    then the x-1, y-1, x+1, y+1 neighbour groups], which fixes both the means and the last-writer of tobj."""
    na, nt = anc.shape[0], targets.shape[0]
    gain = torch.tensor([1., 1., G, G, G, G])
    t = targets * gain
    if nt == 0:
        z = torch.zeros(0, dtype=torch.long)
        return z, z, z, z, torch.zeros(0, 4), torch.zeros(0, 2), z
    r = t[None, :, 4:6] / anc[:, None]
    keep = torch.max(r, 1. / r).max(2)[0] < ANCHOR_T                       # [na, nt]
    a_idx = torch.arange(na).view(na, 1).repeat(1, nt)[keep]
    tt = t.repeat(na, 1, 1)[keep]
    gxy = tt[:, 2:4]
    g = 0.5
    j, k = ((gxy % 1. < g) & (gxy > 1.)).T
    l, m = ((gxy % 1. > (1 - g)) & (gxy < (gain[[2, 3]] - 1.))).T
    off = torch.tensor([[1., 0.], [0., 1.], [-1., 0.], [0., -1.]])
    z = torch.zeros_like(gxy)
    rows_a = torch.cat((a_idx, a_idx[j], a_idx[k], a_idx[l], a_idx[m]), 0)
    rows_t = torch.cat((tt, tt[j], tt[k], tt[l], tt[m]), 0)
    offsets = torch.cat((z, z[j] + off[0], z[k] + off[1], z[l] + off[2], z[m] + off[3]), 0) * g
    b, c = rows_t[:, :2].long().T
    gxy, gwh = rows_t[:, 2:4], rows_t[:, 4:6]
    gij = (gxy - offsets).long()
    gi, gj = gij.T
    return b, rows_a, gj, gi, torch.cat((gxy - gij, gwh), 1), anc[rows_a], c


def focal_bce(pred, true, gamma=1.5, alpha=0.25):
    """losses.py:37-64 around BCEWithLogitsLoss(mean)."""
    loss = F.binary_cross_entropy_with_logits(pred, true, reduction="none")
    p = torch.sigmoid(pred)
    p_t = true * p + (1 - true) * (1 - p)
    loss = loss * ((true * alpha + (1 - true) * (1 - alpha)) * (1.0 - p_t) ** gamma)
    return loss.mean()


def v5_loss(p, targets, anchors, nc, iou_mode="giou"):
    """losses.py:98-156.  p: list of [B, na, G, G, 5+nc]; returns dict(loss, Localization, Classification, Conf_obj)."""
    anc = grid_anchors(anchors)
    lcls, lbox, lobj = torch.zeros(1), torch.zeros(1), torch.zeros(1)
    for i, pi in enumerate(p):
        G = pi.shape[2]
        b, a, gj, gi, tbox, an, tcls = assign_layer(G, targets, anc[i])
        tobj = torch.zeros_like(pi[..., 0])
        if b.shape[0]:
            ps = pi[b, a, gj, gi]
            pxy = ps[:, :2].sigmoid() * 2. - 0.5
            pwh = (ps[:, 2:4].sigmoid() * 2) ** 2 * an
            iou = iou_v5(torch.cat((pxy, pwh), 1).t(), tbox.t(), x1y1x2y2=False, mode=iou_mode)
            lbox = lbox + (1.0 - iou).mean()
            # gr = 1.0.  Duplicate cells: the reference's `tobj[b,a,gj,gi] = ...` is an index_put with repeated
            # indices -- sequential (last row wins) for the sizes its own runs produce, but ATen splits the loop
            # across threads for a few thousand rows and the winner becomes thread-count dependent.  The contract
            # here (and in the HIP kernel) is the deterministic one: the LAST row in row order wins.
            na, G2 = pi.shape[1], G * G
            lin = ((b * na + a) * G + gj) * G + gi
            owner = torch.full((pi.shape[0] * na * G2,), -1, dtype=torch.long)
            owner.scatter_reduce_(0, lin, torch.arange(lin.numel()), reduce="amax", include_self=True)
            hit = owner >= 0
            flat = tobj.view(-1)
            flat[hit] = iou.detach().clamp(0).type(tobj.dtype)[owner[hit]]
            if nc > 1:
                t = torch.zeros_like(ps[:, 5:])
                t[range(b.shape[0]), tcls] = 1.0
                lcls = lcls + focal_bce(ps[:, 5:], t)
        lobj = lobj + focal_bce(pi[..., 4], tobj)
    lbox = lbox * GAIN_BOX
    lobj = lobj * GAIN_OBJ
    lcls = lcls * GAIN_CLS
    return {"loss": lbox + lobj + lcls, "Localization": lbox, "Classification": lcls, "Conf_obj": lobj}
