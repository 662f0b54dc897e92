#!/usr/bin/env python3
"""Golden-vector generator (test infrastructure; runs ONLY in the build container).

Imports the reference's own Python from /root/reference (read-only) under the five
benign shims listed in SURVEY.md section 8c (plus a stub that keeps the pickled colour table from being loaded), feeds it seeded inputs and stores the
inputs + the reference's outputs as small .npz fixtures under tests/golden/.
Nothing of the reference's source travels: fixtures are data only.

    python oracle/gen_golden.py [group ...]        # default: all groups

The GPU box never runs this file (no /root/reference there); tests read the .npz.
"""
import collections
import collections.abc
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, "tests", "golden")
REF = "/root/reference"


def load_reference():
    sys.dont_write_bytecode = True
    sys.modules["cv2"] = types.ModuleType("cv2")                       # shim 1: drawing only
    ti = types.ModuleType("torchinfo"); ti.summary = lambda *a, **k: None
    sys.modules["torchinfo"] = ti                                       # shim 2
    pl = types.ModuleType("pytorch_lightning")
    pl.LightningModule = nn.Module; pl.LightningDataModule = object
    sys.modules["pytorch_lightning"] = pl                               # shim 3
    collections.Iterable = collections.abc.Iterable                     # shim 4
    torch.Tensor.cuda = lambda self, *a, **k: self                      # shim 5: no GPU here
    nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.FloatTensor = torch.FloatTensor
    import pickle                                                       # shim 6: the model modules unpickle a colour table
    pickle.load = lambda f, *a, **k: [(0, 0, 0)] * 256                  # (`dataset/pallete`) at class definition: never loaded
    sys.path.insert(0, REF)
    os.chdir(REF)                                                       # class-level open("dataset//pallete")
    from LightningFunc import accuracy, losses
    return losses, accuracy


def npy(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **{k: npy(v) for k, v in arrays.items()})
    print(f"wrote {path}  ({os.path.getsize(path)} B, {len(arrays)} arrays)")


V4_ANCHORS = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
V4_MASKS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def make_targets(g, B, per_img, C):
    """SURVEY 8d config-2 recipe (draw order matters)."""
    n = B * per_img
    b = torch.arange(B).repeat_interleave(per_img).float()
    cls = torch.randint(0, C, (n,), generator=g).float()
    wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
    xy = torch.rand(n, 2, generator=g) * (1 - wh)
    return torch.cat([b[:, None], cls[:, None], xy, wh], 1)


def assign_cases(G, C):
    """name -> target [nT,6]; B=2.  Anchors are v4 scale-1 anchors scaled to the grid."""
    g = torch.Generator().manual_seed(100 + G)
    plain = make_targets(g, 2, 5, C)
    dup = plain.clone()
    dup[3, 0] = dup[1, 0]; dup[3, 2:6] = dup[1, 2:6] * torch.tensor([1.0, 1.0, 1.02, 0.98])  # same cell+anchor, other box
    dup[3, 1] = (dup[1, 1] + 1) % C                                                         # other label -> multi-hot
    dup[7, 0] = dup[1, 0]; dup[7, 2:6] = dup[1, 2:6]                                         # exact duplicate, third writer
    tie = plain.clone()
    tie[0, 4:6] = torch.tensor([1.0, 1.0]) * 1e-9       # all anchor IoUs ~0 and equal -> first index
    oor = plain.clone(); oor[4, 2] = 1.0                # gi == G -> guard 1 trips for everybody
    oorb = plain.clone(); oorb[2, 0] = 2.0              # b == B
    lab = plain.clone(); lab[6, 1] = float(C)           # guard 2 trips, guard 1 passes
    neg = plain.clone(); neg[5, 1] = -1.0               # label -1 wraps to C-1
    empty = plain[:0].clone()
    edge = plain.clone(); edge[0, 2:4] = torch.tensor([0.0, 0.999999]); edge[1, 2:4] = torch.tensor([0.5, 0.5])
    return dict(plain=plain, dup=dup, tie=tie, oor=oor, oorb=oorb, lab=lab, neg=neg, empty=empty, edge=edge)


def group_yolo_assign(losses, accuracy):
    """G1: build_targets, all ten outputs, edge cases."""
    C, A, B = 4, 3, 2
    arrays = {}
    for G in (13, 26):
        stride = 416 / G
        anchors = torch.FloatTensor([(aw / stride, ah / stride) for aw, ah in V4_ANCHORS[3:6]])
        g = torch.Generator().manual_seed(7 + G)
        pb = torch.rand(B, A, G, G, 4, generator=g) * torch.tensor([G, G, 6.0, 6.0])
        pc = torch.rand(B, A, G, G, C, generator=g)
        arrays[f"G{G}_anchors"] = anchors
        arrays[f"G{G}_pred_boxes"] = pb
        arrays[f"G{G}_pred_cls"] = pc
        for name, tgt in assign_cases(G, C).items():
            outs = accuracy.build_targets(pb, pc, tgt, anchors, 0.5)
            arrays[f"G{G}_{name}_target"] = tgt
            for k, v in zip(("iou_scores", "class_mask", "obj_mask", "noobj_mask", "tx", "ty", "tw", "th", "tcls", "tconf"), outs):
                arrays[f"G{G}_{name}_{k}"] = v
    save("yolo_assign", **arrays)


class _Args:
    def __init__(self, **kw):
        self.__dict__.update(kw)


def group_yolo_loss(losses, accuracy):
    """G2: RegionLoss_v4 / v3 / v2 forward 7 scalars + d loss / d head at small shapes."""
    arrays = {}
    C, B = 4, 2
    for coord in ("smooth_l1_loss", "mse_loss"):
        for cls in ("bce_loss", "ce_loss"):
            args = _Args(model_name="YOLOv4", cls_criterion=cls, coord_criterion=coord)
            crit = losses.configure_loss(args, None, V4_ANCHORS, V4_MASKS, C, 416)
            g = torch.Generator().manual_seed(11)
            heads = [torch.randn(B, 3 * (5 + C), G, G, generator=g).requires_grad_() for G in (52, 26, 13)]
            for case in ("plain", "dup", "oor", "lab", "empty", "neg"):
                tgt = assign_cases(26, C)[case]
                for h in heads:
                    h.grad = None
                out = crit(heads, tgt)
                tag = f"v4_{coord}_{cls}_{case}"
                if case in ("plain", "dup", "lab", "neg", "oor", "empty"):
                    out["loss"].backward()
                for k, v in out.items():
                    arrays[f"{tag}_{k}"] = v
                for i, h in enumerate(heads):
                    arrays[f"{tag}_grad{i}"] = h.grad if h.grad is not None else torch.zeros_like(h)
                arrays[f"{tag}_target"] = tgt
            for i, h in enumerate(heads):
                arrays[f"v4_head{i}"] = h
    # saturated logits: BCE log clamp / backward eps (SURVEY appendix A caveat i)
    args = _Args(model_name="YOLOv4", cls_criterion="bce_loss", coord_criterion="smooth_l1_loss")
    crit = losses.configure_loss(args, None, V4_ANCHORS, V4_MASKS, C, 416)
    g = torch.Generator().manual_seed(12)
    heads = [(torch.randn(B, 3 * (5 + C), G, G, generator=g) * 30).requires_grad_() for G in (52, 26, 13)]
    tgt = assign_cases(26, C)["plain"]
    out = crit(heads, tgt); out["loss"].backward()
    for k, v in out.items():
        arrays[f"v4_sat_{k}"] = v
    for i, h in enumerate(heads):
        arrays[f"v4_sat_head{i}"] = h
        arrays[f"v4_sat_grad{i}"] = h.grad
    arrays["v4_sat_target"] = tgt

    # v3: anchors pre-divided by the stride in the model ctor (YOLOV3.py:54-56), coarse -> fine
    sample = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    v3_anchors = [[(a[0] / s, a[1] / s) for a in sample[i]] for i, s in enumerate((32, 16, 8))]
    args = _Args(model_name="YOLOv3", cls_criterion="bce_loss", coord_criterion="mse_loss")
    crit = losses.configure_loss(args, None, v3_anchors, None, C, 416)
    g = torch.Generator().manual_seed(13)
    heads = [torch.randn(B, 3 * (5 + C), G, G, generator=g).requires_grad_() for G in (13, 26, 52)]
    tgt = assign_cases(13, C)["dup"]
    out = crit(heads, tgt); out["loss"].backward()
    for k, v in out.items():
        arrays[f"v3_{k}"] = v
    for i, h in enumerate(heads):
        arrays[f"v3_head{i}"] = h
        arrays[f"v3_grad{i}"] = h.grad
    arrays["v3_target"] = tgt
    arrays["v3_anchors"] = np.asarray(v3_anchors, dtype=np.float64)

    # v2: 5 anchors in grid units, C=20, config-1 shape [2,125,9,9] at img 288
    v2_anchors = [(1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071)]
    for img in (288, 416):
        G = img // 32
        args = _Args(model_name="YOLOv2", cls_criterion="bce_loss", coord_criterion="smooth_l1_loss")
        crit = losses.configure_loss(args, None, v2_anchors, None, 20, img)
        g = torch.Generator().manual_seed(14 + img)
        head = torch.randn(2, 125, G, G, generator=g).requires_grad_()
        tgt = make_targets(g, 2, 4, 20)
        out = crit(head, tgt); out["loss"].backward()
        for k, v in out.items():
            arrays[f"v2_{img}_{k}"] = v
        arrays[f"v2_{img}_head"] = head
        arrays[f"v2_{img}_grad"] = head.grad
        arrays[f"v2_{img}_target"] = tgt
    save("yolo_loss", **arrays)


def config2_inputs():
    """targets from seed 1 (after x), heads randn seed 2."""
    g = torch.Generator().manual_seed(1)
    _x = torch.rand(16, 3, 416, 416, generator=g)      # drawn first, as in the recipe
    tgt = make_targets(g, 16, 7, 80)
    g2 = torch.Generator().manual_seed(2)
    heads = [torch.randn(16, 255, G, G, generator=g2) for G in (52, 26, 13)]
    return heads, tgt


def group_yolo_config2(losses, accuracy):
    """G3: MultiScaleRegionLoss_v4 at the BASELINE config-2 shape: scalars + gradient digests."""
    heads, tgt = config2_inputs()
    heads = [h.requires_grad_() for h in heads]
    args = _Args(model_name="YOLOv4", cls_criterion="bce_loss", coord_criterion="smooth_l1_loss")
    crit = losses.configure_loss(args, None, V4_ANCHORS, V4_MASKS, 80, 416)
    out = crit(heads, tgt)
    out["loss"].backward()
    arrays = {f"out_{k}": v for k, v in out.items()}
    arrays["target"] = tgt
    for i, h in enumerate(heads):
        gr = h.grad
        arrays[f"head{i}_sum"] = h.detach().double().sum()
        arrays[f"head{i}_first"] = h.detach().flatten()[:64]
        arrays[f"grad{i}_sum"] = gr.double().sum()
        arrays[f"grad{i}_abssum"] = gr.double().abs().sum()
        arrays[f"grad{i}_sqsum"] = (gr.double() ** 2).sum()
        nz = gr.flatten().nonzero().flatten()
        arrays[f"grad{i}_nnz"] = torch.tensor(nz.numel())
        # conf plane of image 0 / anchor 0 and the 64 largest-magnitude entries (index + value)
        G = h.shape[2]
        arrays[f"grad{i}_confplane"] = gr.view(16, 3, 85, G, G)[0, 0, 4]
        top = gr.flatten().abs().topk(64)[1].sort()[0]
        arrays[f"grad{i}_top_idx"] = top
        arrays[f"grad{i}_top_val"] = gr.flatten()[top]
    save("yolo_config2", **arrays)


def group_iou(losses, accuracy):
    """G4: the IoU family on 256 random valid boxes (+ degenerate rows)."""
    g = torch.Generator().manual_seed(21)
    n = 256
    def boxes_xywh(scale):
        wh = torch.rand(n, 2, generator=g) * 0.5 * scale + 0.01 * scale
        xy = torch.rand(n, 2, generator=g) * scale
        return torch.cat([xy, wh], 1)
    a, b = boxes_xywh(416.0), boxes_xywh(416.0)
    b[:32] = a[:32] * (1 + 0.1 * torch.rand(32, 4, generator=g))   # overlapping pairs
    b[32:40] = a[32:40]                                            # identical
    def to_xyxy(x):
        return torch.cat([x[:, :2] - x[:, 2:] / 2, x[:, :2] + x[:, 2:] / 2], 1)
    ax, bx = to_xyxy(a), to_xyxy(b)
    arrays = dict(a_xywh=a, b_xywh=b, a_xyxy=ax, b_xyxy=bx)
    arrays["bbox_iou_xyxy"] = accuracy.bbox_iou(ax, bx, x1y1x2y2=True)
    arrays["bbox_iou_xywh"] = accuracy.bbox_iou(a, b, x1y1x2y2=False)
    arrays["bbox_iou_bcast"] = accuracy.bbox_iou(ax[0:1], bx, x1y1x2y2=True)
    for mode, kw in (("iou", {}), ("giou", dict(GIoU=True)), ("diou", dict(DIoU=True)), ("ciou", dict(CIoU=True))):
        arrays[f"v5_{mode}_xywh"] = accuracy.bbox_iou_v5(a.t(), b.t(), x1y1x2y2=False, **kw)
        arrays[f"v5_{mode}_xyxy"] = accuracy.bbox_iou_v5(ax.t(), bx.t(), x1y1x2y2=True, **kw)
    anchors = torch.FloatTensor([(36 / 16, 75 / 16), (76 / 16, 55 / 16), (72 / 16, 146 / 16)])
    gwh = a[:, 2:] / 16
    arrays["wh_anchors"] = anchors
    arrays["wh_gwh"] = gwh
    arrays["wh_iou"] = torch.stack([accuracy.bbox_wh_iou(an, gwh) for an in anchors])
    an = ax / 416.0; bn = bx / 416.0
    arrays["corner_a"] = an; arrays["corner_b"] = bn
    arrays["corner_iou"] = accuracy.iou(an, bn)
    rl = losses.RetinaNetLoss(None, losses.focal_loss, nn.SmoothL1Loss, 4, 600)
    arrays["retina_box_iou_xywh"] = rl.box_iou(a[:64], b[:16], order="xywh")
    arrays["retina_box_iou_xyxy"] = rl.box_iou(ax[:64], bx[:16], order="xyxy")
    save("iou", **arrays)


def load_retina_utils():
    from LightningFunc.utils import RetinaUtils, SSDUtils
    return RetinaUtils, SSDUtils


def group_anchors(losses, accuracy):
    """G5: anchor / prior tables (bit-exact targets) + YOLO scaled anchors."""
    import hashlib
    RetinaUtils, SSDUtils = load_retina_utils()
    arrays = {}
    for w, h in ((600, 600), (1333, 800), (128, 128), (96, 160)):
        t = RetinaUtils.get_anchor_boxes(torch.Tensor([w, h]))
        tag = f"retina_{w}x{h}"
        arrays[tag + "_n"] = torch.tensor(t.shape[0])
        arrays[tag + "_sha256"] = np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8)
        arrays[tag + "_head"] = t[:64]
        arrays[tag + "_tail"] = t[-64:]
        if w <= 160:
            arrays[tag + "_full"] = t
    arrays["retina_wh_table"] = RetinaUtils.get_anchor_wh([32*32., 64*64., 128*128., 256*256., 512*512.], [1/2., 1/1., 2/1.], [1., pow(2,1/3.), pow(2,2/3.)])
    arrays["ssd_dboxes"] = SSDUtils.get_dboxes()
    for name, anchors, img in (("v4s0", V4_ANCHORS[0:3], 416), ("v4s1", V4_ANCHORS[3:6], 416), ("v4s2", V4_ANCHORS[6:9], 416),
                               ("v2", [(1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071)], 416)):
        for G in (9, 13, 26, 52):
            stride = img / G
            arrays[f"yolo_{name}_G{G}"] = torch.FloatTensor([(aw / stride, ah / stride) for aw, ah in anchors])
    save("anchors", **arrays)


def retina_inputs(seed, B, C, W, H, n_gt, scale=0.3):
    RetinaUtils, _ = load_retina_utils()
    g = torch.Generator().manual_seed(seed)
    anchors = RetinaUtils.get_anchor_boxes(torch.Tensor([W, H]))
    An = anchors.shape[0]
    loc = torch.randn(B, An, 4, generator=g) * scale
    cls = torch.randn(B, An, C, generator=g)
    n = B * n_gt
    b = torch.arange(B).repeat_interleave(n_gt).float()
    lab = torch.randint(0, C, (n,), generator=g).float()
    wh = torch.rand(n, 2, generator=g) * 0.45 + 0.25
    xy = torch.rand(n, 2, generator=g) * 0.6 + 0.2          # centres (the loss reads x,y as centres)
    tgt = torch.cat([b[:, None], lab[:, None], xy, wh], 1)
    return anchors, loc, cls, tgt


def group_retina(losses, accuracy):
    """G6: RetinaNetLoss (+ focal_loss alone) forward and gradients at 128x128 anchors."""
    arrays = {}
    C = 4
    for coord_name, coord in (("smooth_l1_loss", nn.SmoothL1Loss), ("mse_loss", nn.MSELoss)):
        anchors, loc, cls, tgt = retina_inputs(31, 2, C, 128, 128, 3)
        loc.requires_grad_(); cls.requires_grad_()
        crit = losses.RetinaNetLoss(anchors, losses.focal_loss, coord, C, 128)
        out = crit((loc, cls), tgt)
        out["loss"].backward()
        tag = f"r128_{coord_name}"
        for k, v in out.items():
            arrays[f"{tag}_{k}"] = v
        arrays[f"{tag}_grad_loc"] = loc.grad
        arrays[f"{tag}_grad_cls"] = cls.grad
    arrays["r128_anchors"] = anchors; arrays["r128_loc"] = loc; arrays["r128_cls"] = cls; arrays["r128_target"] = tgt
    # the encoded targets themselves (index work)
    crit = losses.RetinaNetLoss(anchors, losses.focal_loss, nn.SmoothL1Loss, C, 128)
    for bid in range(2):
        boxes = tgt[tgt[:, 0] == bid][:, 2:] * 128
        ious = crit.box_iou(anchors, boxes, order="xywh")
        mi, mid = ious.max(1)
        arrays[f"r128_maxiou_{bid}"] = mi; arrays[f"r128_maxid_{bid}"] = mid
    # saturated / clamped logits for the softmax clamp(1e-7, 1-1e-7) branch
    anchors, loc, cls, tgt = retina_inputs(32, 2, C, 96, 160, 2)
    cls = (cls * 12).requires_grad_(); loc.requires_grad_()
    crit = losses.RetinaNetLoss(anchors, losses.focal_loss, nn.SmoothL1Loss, C, 128)
    out = crit((loc, cls), tgt); out["loss"].backward()
    for k, v in out.items():
        arrays[f"rsat_{k}"] = v
    arrays["rsat_anchors"] = anchors; arrays["rsat_loc"] = loc; arrays["rsat_cls"] = cls; arrays["rsat_target"] = tgt
    arrays["rsat_grad_loc"] = loc.grad; arrays["rsat_grad_cls"] = cls.grad
    # focal_loss alone
    g = torch.Generator().manual_seed(33)
    x = (torch.randn(64, 6, generator=g) * 3).requires_grad_()
    y = torch.randint(0, 7, (64,), generator=g)
    fl = losses.focal_loss(6, reduction="sum")(x, y); fl.backward()
    arrays["focal_x"] = x; arrays["focal_y"] = y; arrays["focal_sum"] = fl; arrays["focal_grad"] = x.grad
    arrays["focal_none"] = losses.focal_loss(6, reduction="none")(x.detach(), y)
    save("retina", **arrays)


def nms_heads(seed, B, C, grids, spread):
    """Heads whose first five channels are valid geometry: centres in clusters, positive sizes, raw conf
    around the -0.0151 threshold; no exact score ties."""
    g = torch.Generator().manual_seed(seed)
    heads = []
    for G in grids:
        h = torch.randn(B, 3, 5 + C, G, G, generator=g)
        centres = torch.rand(B, 3, 2, 1, 1, generator=g) * 300 + 50
        h[:, :, 0:2] = centres + torch.randn(B, 3, 2, G, G, generator=g) * spread
        h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 80 + 20
        h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.8 + 0.4
        h[:, :, 5:] = torch.rand(B, 3, C, G, G, generator=g)
        heads.append(h.view(B, 3 * (5 + C), G, G).contiguous())
    return heads


def group_nms(losses, accuracy):
    """G7: YOLO weighted-merge NMS (v4 / v2-style 5 anchors is the same code / v5 reinterpretation) and the
    Retina/SSD greedy top-k NMS incl. the 'one box left' and 'nothing above threshold' cases."""
    from model.YOLOV4 import YOLOv4
    from model.YOLOV5 import YOLOv5
    from model.RetinaNet import RetinaNet
    RetinaUtils, _ = load_retina_utils()
    arrays = {}
    for name, seed, grids, spread in (("a", 51, (8, 4, 2), 25.0), ("b", 52, (13, 7), 60.0), ("c", 53, (6,), 5.0)):
        heads = nms_heads(seed, 2, 4, grids, spread)
        if name == "c":
            heads[0][1, :, :, :] = heads[0][1]          # image 1: everything below the conf threshold
            hv = heads[0].view(2, 3, 9, 6, 6); hv[1, :, 4] = -1.0
        out = YOLOv4.non_max_suppression(None, [h.clone() for h in heads])
        for i, h in enumerate(heads):
            arrays[f"yolo_{name}_head{i}"] = h
        for b, o in enumerate(out):
            arrays[f"yolo_{name}_out{b}"] = o if o is not None else torch.zeros(0, 7)
            arrays[f"yolo_{name}_none{b}"] = torch.tensor(o is None)
    # v5: [B,3,G,G,no] memory reinterpreted as [B,3,no,G,G] (YOLOV5.py:176-182)
    heads = nms_heads(54, 2, 4, (8, 4), 25.0)
    v5 = [h.view(2, 3, 8, 8, 9) if h.shape[2] == 8 else h.view(2, 3, 4, 4, 9) for h in heads]
    out = YOLOv5.non_max_suppression(None, [h.clone() for h in v5])
    for i, h in enumerate(v5):
        arrays[f"v5_head{i}"] = h
    for b, o in enumerate(out):
        arrays[f"v5_out{b}"] = o
    # Retina / SSD greedy
    class _M:                       # the method only reads self.iou_boxes
        pass
    m = _M()
    m.iou_boxes = RetinaUtils.get_anchor_boxes(torch.Tensor([128, 128]))
    An = m.iou_boxes.shape[0]
    g = torch.Generator().manual_seed(55)
    for name, shift in (("typ", -1.6), ("few", -3.2), ("none", -9.0), ("many", 0.5)):
        loc = torch.randn(2, An, 4, generator=g) * 0.2
        cls = torch.randn(2, An, 4, generator=g) + shift
        if name == "few":
            cls[:] = -9.0
            cls[0, 100, 1] = 2.0; cls[0, 101, 2] = 1.5; cls[0, 2000, 0] = 1.0      # 3 candidates in image 0
            cls[1, 7, 3] = 3.0; cls[1, 1500, 3] = 2.5                                 # 2 candidates in image 1
        out = RetinaNet.non_max_suppression(m, (loc, cls))
        arrays[f"ret_{name}_loc"] = loc; arrays[f"ret_{name}_cls"] = cls
        for b, o in enumerate(out):
            arrays[f"ret_{name}_out{b}"] = o
    arrays["ret_anchors"] = m.iou_boxes
    save("nms", **arrays)


def group_decode(losses, accuracy):
    """G9: yolo_forward_dynamic / YoloLayer / get_region_boxes."""
    from LightningFunc.utils.YoloV4Utils import get_region_boxes, yolo_forward_dynamic
    from model.YOLOV4 import YoloLayer
    arrays = {}
    A18 = [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401]
    g = torch.Generator().manual_seed(71)
    C, B = 5, 2
    heads = [torch.randn(B, 3 * (5 + C), G, G, generator=g) * 1.5 for G in (13, 7, 4)]
    layers = [YoloLayer(anchor_mask=m, num_classes=C, anchors=A18, num_anchors=9, stride=s)
              for m, s in (([0, 1, 2], 8), ([3, 4, 5], 16), ([6, 7, 8], 32))]
    outs = []
    for l, h in zip(layers, heads):
        l.eval()
        outs.append(l(h))
    boxes, confs = get_region_boxes(outs)
    for i, h in enumerate(heads):
        arrays[f"head{i}"] = h
    arrays["boxes"] = boxes; arrays["confs"] = confs
    # non-square, scale_x_y != 1
    h = torch.randn(1, 2 * (5 + 3), 5, 9, generator=g)
    b, c = yolo_forward_dynamic(h, 0.6, 3, [1.5, 2.0, 3.0, 1.0], 2, scale_x_y=1.2)
    arrays["ns_head"] = h; arrays["ns_boxes"] = b; arrays["ns_confs"] = c
    save("decode", **arrays)


V5_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def group_yolov5(losses, accuracy):
    """G6b: build_targets_v5 rows and MultiScaleRegionLoss_v5 forward + gradients (GIoU, as the reference calls it)."""
    arrays = {}
    C, B = 4, 2
    args = _Args(model_name="YOLOv5", cls_criterion="bce_loss", coord_criterion="smooth_l1_loss", type="Yolov5s")
    crit = losses.configure_loss(args, None, V5_ANCHORS, None, C, 640)
    g = torch.Generator().manual_seed(81)
    for case in ("plain", "dup", "empty"):
        heads = [torch.randn(B, 3, G, G, 5 + C, generator=g).requires_grad_() for G in (16, 8, 4)]
        n = 0 if case == "empty" else 6
        wh = torch.rand(n, 2, generator=g) * 0.5 + 0.03
        tgt = torch.cat([torch.randint(0, B, (n, 1), generator=g).float(), torch.randint(0, C, (n, 1), generator=g).float(),
                         torch.rand(n, 2, generator=g) * 0.9 + 0.05, wh], 1)
        if case == "dup":
            tgt[3] = tgt[1]; tgt[3, 1] = (tgt[1, 1] + 1) % C; tgt[3, 4:6] = tgt[1, 4:6] * 1.1     # same cells, other box/label
        out = crit(heads, tgt)
        out["loss"].backward()
        for k, v in out.items():
            arrays[f"{case}_{k}"] = v
        arrays[f"{case}_target"] = tgt
        for i, h in enumerate(heads):
            arrays[f"{case}_head{i}"] = h
            arrays[f"{case}_grad{i}"] = h.grad if h.grad is not None else torch.zeros_like(h)
        tcls, tbox, indices, anch = accuracy.build_targets_v5(heads, tgt, crit.anchors, 3, 3)
        for i in range(3):
            arrays[f"{case}_L{i}_b"], arrays[f"{case}_L{i}_a"], arrays[f"{case}_L{i}_gj"], arrays[f"{case}_L{i}_gi"] = indices[i]
            arrays[f"{case}_L{i}_tbox"] = tbox[i]; arrays[f"{case}_L{i}_anch"] = anch[i]; arrays[f"{case}_L{i}_tcls"] = tcls[i]
    arrays["grid_anchors"] = crit.anchors
    save("yolov5", **arrays)


def group_ssd(losses, accuracy):
    """G6c: SSDLoss with ce_loss and focal_loss (+ both coord criteria) on the real 8 732-prior table."""
    _, SSDUtils = load_retina_utils()
    priors = SSDUtils.get_dboxes()
    arrays = {"priors": priors}
    C, B = 6, 3
    g = torch.Generator().manual_seed(95)
    off = (torch.randn(B, 8732, 4, generator=g) * 0.5)
    cls = torch.randn(B, 8732, C, generator=g)
    n = 7
    wh = torch.rand(n, 2, generator=g) * 0.4 + 0.1
    tgt = torch.cat([torch.tensor([0., 0, 0, 1, 1, 2, 2])[:, None], torch.randint(1, C, (n, 1), generator=g).float(),
                     torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1)
    arrays["off"] = off; arrays["cls"] = cls; arrays["target"] = tgt
    tgt2 = tgt.clone(); tgt2[:, 0] = torch.tensor([0., 0, 0, 0, 2, 2, 2])        # only 2 distinct image ids for B = 3
    arrays["target2"] = tgt2
    for cname, ccls in (("ce_loss", nn.CrossEntropyLoss), ("focal_loss", losses.focal_loss)):
        for kname, kcls in (("smooth_l1_loss", nn.SmoothL1Loss), ("mse_loss", nn.MSELoss)):
            for tname, tt in (("t1", tgt), ("t2", tgt2)):
                crit = losses.SSDLoss(priors, ccls, kcls, C, 300)
                o = off.clone().requires_grad_(); c = cls.clone().requires_grad_()
                out = crit((o, c), tt)
                out["loss"].backward()
                tag = f"{cname}_{kname}_{tname}"
                for k, v in out.items():
                    arrays[f"{tag}_{k}"] = v
                arrays[f"{tag}_grad_off"] = o.grad; arrays[f"{tag}_grad_cls"] = c.grad
    crit = losses.SSDLoss(priors, nn.CrossEntropyLoss, nn.SmoothL1Loss, C, 300)
    bwa, mb = crit.match(priors, tgt[:3, 2:], 0.5)
    arrays["match_ann"] = bwa; arrays["match_bin"] = mb
    save("ssd", **arrays)


def group_metrics(losses, accuracy):
    """G8: get_batch_statistics, ap_per_class, compute_ap."""
    arrays = {}
    g = torch.Generator().manual_seed(97)
    B = 3
    outs, n_out = [], [40, 0, 25]
    tg = []
    for b in range(B):
        nt = [6, 3, 4][b]
        xy = torch.rand(nt, 2, generator=g) * 300
        wh = torch.rand(nt, 2, generator=g) * 80 + 20
        lab = torch.randint(0, 4, (nt, 1), generator=g).float()
        tg.append(torch.cat([torch.full((nt, 1), float(b)), lab, xy, xy + wh], 1))     # used as corners (quirk 6)
        if n_out[b] == 0:
            outs.append(None)
            continue
        k = n_out[b]
        src = tg[-1][torch.randint(0, nt, (k,), generator=g)]
        box = src[:, 2:6] + torch.randn(k, 4, generator=g) * 12
        conf = torch.rand(k, 1, generator=g)
        lbl = torch.where(torch.rand(k, 1, generator=g) < 0.7, src[:, 1:2], torch.randint(0, 5, (k, 1), generator=g).float())
        outs.append(torch.cat([box, conf, torch.rand(k, 1, generator=g), lbl], 1))
    targets = torch.cat(tg, 0)
    stats = accuracy.get_batch_statistics(outs, targets, 0.5)
    arrays["targets"] = targets
    for b, o in enumerate(outs):
        arrays[f"out{b}"] = o if o is not None else torch.zeros(0, 7)
    for i, (tp, sc, lb) in enumerate(stats):
        arrays[f"stat{i}_tp"] = tp; arrays[f"stat{i}_score"] = sc; arrays[f"stat{i}_label"] = lb
    arrays["nstats"] = np.int64(len(stats))
    tp = np.concatenate([s[0] for s in stats]); conf = np.concatenate([s[1] for s in stats]); pc = np.concatenate([s[2] for s in stats])
    tcls = targets[:, 1].numpy()
    p, r, ap, f1, uc = accuracy.ap_per_class(tp, conf, pc, tcls)
    arrays.update(ap_tp=tp, ap_conf=conf, ap_pred_cls=pc, ap_target_cls=tcls, ap_p=p, ap_r=r, ap_ap=ap, ap_f1=f1, ap_classes=uc)
    # a larger random case
    rng = np.random.RandomState(5)
    n = 5000
    conf2 = rng.rand(n).astype(np.float32); pc2 = rng.randint(0, 12, n).astype(np.float32)
    tp2 = (rng.rand(n) < conf2 * 0.8).astype(np.float64); tc2 = rng.randint(0, 10, 900).astype(np.float32)
    p, r, ap, f1, uc = accuracy.ap_per_class(tp2, conf2, pc2, tc2)
    arrays.update(big_tp=tp2, big_conf=conf2, big_pred_cls=pc2, big_target_cls=tc2, big_p=p, big_r=r, big_ap=ap, big_f1=f1, big_classes=uc)
    rec = np.array([0.1, 0.1, 0.3, 0.5, 0.5, 0.9]); prec = np.array([1.0, 0.5, 0.66, 0.75, 0.6, 0.5])
    arrays["cap_recall"] = rec; arrays["cap_precision"] = prec; arrays["cap_ap"] = accuracy.compute_ap(rec, prec)
    save("metrics", **arrays)


def group_yolo_stats(losses, accuracy):
    """G8b: get_yolo_statistics (accuracy.py:382-470) on a 2-scale YOLOv4-style model stub."""
    torch.cuda.FloatTensor = torch.FloatTensor
    class _M:
        pass
    m = _M()
    m.anchors = V4_ANCHORS; m.anch_masks = [[3, 4, 5], [6, 7, 8]]; m.num_classes = 4; m.img_size = 416; m.ignore_thres = 0.5
    g = torch.Generator().manual_seed(99)
    heads = [torch.randn(2, 27, G, G, generator=g) for G in (26, 13)]
    tgt = assign_cases(26, 4)["dup"]
    out = accuracy.get_yolo_statistics(m, heads, tgt)
    arrays = {"target": tgt}
    for i, h in enumerate(heads):
        arrays[f"head{i}"] = h
    for G, vals in out.items():
        for name, v in zip(("cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj", "output"), vals):
            arrays[f"G{G}_{name}"] = v
    save("yolo_stats", **arrays)


GROUPS = {
    "yolo_assign": group_yolo_assign,
    "yolo_loss": group_yolo_loss,
    "yolo_config2": group_yolo_config2,
    "iou": group_iou,
    "anchors": group_anchors,
    "retina": group_retina,
    "nms": group_nms,
    "decode": group_decode,
    "yolov5": group_yolov5,
    "ssd": group_ssd,
    "metrics": group_metrics,
    "yolo_stats": group_yolo_stats,
}


# ---- caller models: state_dict manifests + forward fingerprints under a seeded deterministic fill -------------------
def seeded_fill(model, seed):
    """Deterministic parameters/buffers from the state_dict KEY ORDER and shapes alone, so that two implementations with
    the same manifest get the same weights: 4-D conv weights ~ U(-1, 1) / sqrt(fan_in), 1-D `weight` (BN gamma) in
    [0.75, 1.25], `bias` / `running_mean` in [-0.1, 0.1], `running_var` in [0.75, 1.25], integer buffers untouched."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for k, v in model.state_dict().items():
            if not v.is_floating_point():
                continue
            r = torch.rand(v.shape, generator=g)
            if v.dim() >= 2:
                fan_in = v[0].numel()
                v.copy_((r * 2 - 1) / fan_in ** 0.5)
            elif k.endswith("running_var") or k.endswith("weight"):
                v.copy_(r * 0.5 + 0.75)
            else:
                v.copy_(r * 0.2 - 0.1)


def fingerprint(outs):
    """Per output tensor: shape, sum, sum of |.|, and 64 evenly strided samples (enough to catch any wiring difference)."""
    fp = {}
    for i, t in enumerate(outs):
        t = t.detach().float().contiguous()
        flat = t.flatten()
        step = max(1, flat.numel() // 64)
        fp[f"out{i}_shape"] = np.array(t.shape, dtype=np.int64)
        fp[f"out{i}_sum"] = np.array([flat.double().sum().item(), flat.double().abs().sum().item()])
        fp[f"out{i}_samples"] = npy(flat[::step][:64])
    return fp


MODEL_CASES = {            # name -> (module, class, classes, args overrides, input size)
    "yolov4": ("model.YOLOV4", "YOLOv4", 80, dict(model_name="YOLOv4"), 64),
    "yolov2": ("model.YOLOV2", "YOLOv2", 20, dict(model_name="YOLOv2"), 64),
    "yolov3": ("model.YOLOV3", "YOLOv3", 80, dict(model_name="YOLOv3", coord_criterion="mse_loss"), 64),
    "yolov5s": ("model.YOLOV5", "YOLOv5", 80, dict(model_name="YOLOv5", type="Yolov5s"), 64),
    "yolov5l": ("model.YOLOV5", "YOLOv5", 80, dict(model_name="YOLOv5", type="Yolov5l"), 64),
    "retinanet": ("model.RetinaNet", "RetinaNet", 4, dict(model_name="RetinaNet", cls_criterion="focal_loss"), 64),
}


def model_input(size, batch):
    return torch.rand(batch, 3, size, size, generator=torch.Generator().manual_seed(77))


def group_models(losses, accuracy):
    """state_dict key/shape manifests (tests/golden/state_dict_*.json) and forward fingerprints (models_forward.npz) of the
    reference's model classes.  The reference unpickles `dataset/pallete` at class-definition time (a colour table for the
    drawing helpers); that load is NOT performed -- `pickle.load` is stubbed in `load_reference` (shim 6)."""
    import importlib
    import json
    mods = {m: importlib.import_module(m) for m in sorted({c[0] for c in MODEL_CASES.values()})}
    fps, v5_manifest = {}, {}
    for name, (mod, cls, nc, over, size) in MODEL_CASES.items():
        args = _Args(**{**dict(cls_criterion="bce_loss", coord_criterion="smooth_l1_loss", data_module="x"), **over})
        torch.manual_seed(0)
        model = getattr(mods[mod], cls)([str(i) for i in range(nc)], args)
        manifest = {k: list(v.shape) for k, v in model.state_dict().items()}
        if name.startswith("yolov5"):
            v5_manifest[over["type"]] = manifest
        else:
            with open(os.path.join(OUT, f"state_dict_{name}.json"), "w") as f:
                json.dump(manifest, f)
        seeded_fill(model, 1234)
        with torch.no_grad():
            model.eval()
            o = model(model_input(size, 1))
            for k, v in fingerprint(list(o) if isinstance(o, (list, tuple)) else [o]).items():
                fps[f"{name}_eval_{k}"] = v
            model.train()
            o = model(model_input(size, 2))
            for k, v in fingerprint(list(o) if isinstance(o, (list, tuple)) else [o]).items():
                fps[f"{name}_train_{k}"] = v
        print(name, "ok", len(manifest), "keys")
    with open(os.path.join(OUT, "state_dict_yolov5.json"), "w") as f:
        json.dump(v5_manifest, f)
    save("models_forward", **fps)


GROUPS["models"] = group_models


def main(argv):
    names = argv or list(GROUPS)
    losses, accuracy = load_reference()
    torch.set_num_threads(8)
    for n in names:
        GROUPS[n](losses, accuracy)


if __name__ == "__main__":
    main(sys.argv[1:])
