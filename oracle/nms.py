"""Oracle (test infrastructure): the YOLO confidence-weighted merge NMS, restated on torch-CPU from
`model/YOLOV4.py:221-283` (== YOLOV3.py:273-335, YOLOV2.py:159-222, YOLOV5.py:157-218 on reinterpreted memory).
The Retina/SSD greedy NMS lives in oracle/retina.py.  Pinned by tests/golden/nms.npz."""
import torch

from .boxes import iou_plus1, xywh2xyxy

CONF_THRES = -0.0151        # YOLOV4.py:228: the argument is overwritten


def flatten_heads(heads, num_anchors=3):
    """[B, A*(5+C), G, G] (raw logits, NO sigmoid / grid decode: YOLOV4.py:126-134,160) -> [B, sum A*G*G, 5+C]
    in (anchor, row, col) order per scale, scales concatenated (YOLOV4.py:237-251)."""
    rows = []
    for h in heads:
        B, ch, G, _ = h.shape
        ans = ch // num_anchors
        rows.append(h.view(B, num_anchors, ans, G, G).permute(0, 1, 3, 4, 2).reshape(B, -1, ans))
    return torch.cat(rows, 1)


def yolo_weighted_nms(heads, nms_thres=0.4, num_anchors=3):
    """Returns a list (per image) of [K,7] tensors (x1,y1,x2,y2,conf,cls_conf,cls_pred) or None.
    One deliberate difference: the pivot is always retired, so the loop terminates even when a box's
    self-IoU is <= nms_thres (the reference spins forever there, SURVEY section 7 hard part 6)."""
    pred = flatten_heads(heads, num_anchors).clone()
    pred[..., :4] = xywh2xyxy(pred[..., :4])
    out = []
    for img in pred:
        img = img[img[:, 4] >= CONF_THRES]
        if not img.size(0):
            out.append(None)
            continue
        score = img[:, 4] * img[:, 5:].max(1)[0]
        img = img[(-score).argsort()]
        cconf, cpred = img[:, 5:].max(1, keepdim=True)
        det = torch.cat((img[:, :5], cconf.float(), cpred.float()), 1)
        alive = torch.ones(det.size(0), dtype=torch.bool)
        keep = []
        while alive.any():
            idx = alive.nonzero().flatten()
            cur = det[idx]
            inv = (iou_plus1(cur[0:1, :4], cur[:, :4]) > nms_thres) & (cur[0, -1] == cur[:, -1])
            w = cur[inv, 4:5]
            top = cur[0].clone()
            top[:4] = (w * cur[inv, :4]).sum(0) / w.sum()
            keep.append(top)
            dead = idx[inv]
            alive[dead] = False
            alive[idx[0]] = False
        out.append(torch.stack(keep))
    return out
