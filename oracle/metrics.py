"""Oracle (test infrastructure): evaluation statistics restated from `LightningFunc/accuracy.py:116-154`
(get_batch_statistics) and `207-287` (ap_per_class, compute_ap).  Pinned by tests/golden/metrics.npz."""
import numpy as np
import torch

from .boxes import iou_plus1


def batch_statistics(outputs, targets, iou_threshold):
    """accuracy.py:116-154.  Predictions (x1,y1,x2,y2,conf,cls_conf,label) are matched in their given order;
    target columns 2:6 are read as corners (quirk 6).  -> list of [tp f64[K], scores, labels] for non-None outputs."""
    res = []
    for i, out in enumerate(outputs):
        if out is None:
            continue
        ann = targets[targets[:, 0] == i][:, 1:]
        tp = np.zeros(out.shape[0])
        if len(ann):
            labels, boxes, used = ann[:, 0], ann[:, 1:], []
            for k in range(out.shape[0]):
                if len(used) == len(ann):
                    break
                if not bool((labels == out[k, -1]).any()):
                    continue
                v, j = iou_plus1(out[k:k + 1, :4], boxes).max(0)
                if v >= iou_threshold and int(j) not in used:
                    tp[k] = 1
                    used.append(int(j))
        res.append([tp, out[:, 4].numpy(), out[:, -1].numpy()])
    return res


def compute_ap(recall, precision):
    """accuracy.py:262-287: precision envelope, area where recall changes."""
    mrec = np.concatenate(([0.0], recall, [1.0]))
    mpre = np.concatenate(([0.0], precision, [0.0]))
    mpre = np.maximum.accumulate(mpre[::-1])[::-1]
    i = np.where(mrec[1:] != mrec[:-1])[0]
    return np.sum((mrec[i + 1] - mrec[i]) * mpre[i + 1])


def ap_per_class(tp, conf, pred_cls, target_cls):
    """accuracy.py:207-260."""
    order = np.argsort(-conf)
    tp, conf, pred_cls = tp[order], conf[order], pred_cls[order]
    classes = np.unique(target_cls)
    ap, p, r = [], [], []
    for c in classes:
        sel = pred_cls == c
        n_gt, n_p = (target_cls == c).sum(), sel.sum()
        if n_p == 0 and n_gt == 0:
            continue
        if n_p == 0 or n_gt == 0:
            ap.append(0); r.append(0); p.append(0)
            continue
        fpc, tpc = (1 - tp[sel]).cumsum(), tp[sel].cumsum()
        rc, pc = tpc / (n_gt + 1e-16), tpc / (tpc + fpc)
        r.append(rc[-1]); p.append(pc[-1]); ap.append(compute_ap(rc, pc))
    p, r, ap = np.array(p), np.array(r), np.array(ap)
    return p, r, ap, 2 * p * r / (p + r + 1e-16), classes.astype("int32")
