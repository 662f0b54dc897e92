"""Host-side mirror of the reference's `LightningFunc/accuracy.py` hot-path functions; each is one
C-ABI call into libodpl_hip.so.  (Drawing helpers mark_target/mark_pred are visualisation: out of scope.)

    build_targets      accuracy.py:305-380
"""
import torch

from .. import _lib


def build_targets(pred_boxes, pred_cls, target, anchors, ignore_thres):
    """accuracy.py:305-380.  Same argument meaning and the same ten outputs, same dtypes
    (obj_mask / noobj_mask uint8, the rest fp32).  Out-of-range labels/coordinates are not errors:
    they trip the reference's all-or-nothing guards."""
    _lib.require_cuda(pred_boxes, pred_cls, target)
    L = _lib.lib()
    dev = pred_boxes.device
    nB, nA, nG = pred_boxes.size(0), pred_boxes.size(1), pred_boxes.size(2)
    nC = pred_cls.size(-1)
    pb = pred_boxes.float().contiguous()
    pc = pred_cls.float().contiguous()
    tg = target.float().contiguous()
    an = torch.as_tensor(anchors, dtype=torch.float32, device=dev).contiguous()
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    shape = (nB, nA, nG, nG)
    state = torch.empty(shape, dtype=torch.int32, device=dev)
    obj_mask = torch.empty(shape, dtype=torch.uint8, device=dev)
    noobj_mask = torch.empty(shape, dtype=torch.uint8, device=dev)
    iou_scores, class_mask, tx, ty, tw, th, tconf = (f(*shape) for _ in range(7))
    tcls = f(*shape, nC)
    status = torch.empty(8, dtype=torch.int32, device=dev)
    _lib.check(L.odpl_yolo_build_targets(
        _lib.ptr(pb), _lib.ptr(pc), _lib.ptr(tg) if tg.shape[0] else None, _lib.ptr(an),
        nB, nA, nG, nC, tg.shape[0], float(ignore_thres), _lib.ptr(state),
        _lib.ptr(iou_scores), _lib.ptr(class_mask), _lib.ptr(obj_mask), _lib.ptr(noobj_mask),
        _lib.ptr(tx), _lib.ptr(ty), _lib.ptr(tw), _lib.ptr(th), _lib.ptr(tcls), _lib.ptr(tconf),
        _lib.ptr(status), _lib.current_stream(dev)), "odpl_yolo_build_targets")
    return iou_scores, class_mask, obj_mask, noobj_mask, tx, ty, tw, th, tcls, tconf
