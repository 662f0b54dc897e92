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


def _f32c(t):
    return t.float().contiguous()


def _pair(box1, box2, kind, x1y1x2y2, layout_4xn):
    _lib.require_cuda(box1, box2)
    b1, b2 = _f32c(box1), _f32c(box2)
    n = b2.shape[1] if layout_4xn else b2.shape[0]
    n1 = b1.shape[1] if layout_4xn else b1.shape[0]
    if layout_4xn and b1.dim() == 1:
        b1 = b1.view(4, 1); n1 = 1
    out = torch.empty(n, dtype=torch.float32, device=b2.device)
    _lib.check(_lib.lib().odpl_pair_iou(_lib.ptr(b1), _lib.ptr(b2), _lib.ptr(out), n, n1, kind, 1 if x1y1x2y2 else 0,
                                        1 if layout_4xn else 0, _lib.current_stream(b2.device)), "odpl_pair_iou")
    return out


def bbox_iou(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False):
    """accuracy.py:39-69: "+1" pixel IoU of [N,4] vs [N,4] (or [1,4] broadcast).  The reference accepts the
    GIoU/DIoU/CIoU keywords and ignores them; so does this."""
    return _pair(box1, box2, 0, x1y1x2y2, False)


def bbox_iou_v5(box1, box2, x1y1x2y2=True, GIoU=False, DIoU=False, CIoU=False):
    """accuracy.py:71-114: transposed layout ([4,N]); GIoU takes precedence, then DIoU, then CIoU.  Forward only
    (the differentiable use is inside MultiScaleRegionLoss_v5)."""
    kind = 2 if GIoU else 3 if DIoU else 4 if CIoU else 1
    return _pair(box1, box2, kind, x1y1x2y2, True)


def iou(tens1, tens2):
    """accuracy.py:6-37: corner IoU without +1 / epsilon on equal-shaped [..., 4] tensors."""
    assert tens1.size() == tens2.size() and tens1.size(-1) == 4
    shp = tens1.shape[:-1]
    return _pair(tens1.reshape(-1, 4), tens2.reshape(-1, 4), 5, True, False).view(shp)


def bbox_wh_iou(wh1, wh2):
    """accuracy.py:297-303: one anchor (w,h) against gwh [nT,2] -> [nT]."""
    _lib.require_cuda(wh2)
    a = torch.as_tensor(wh1, dtype=torch.float32, device=wh2.device).reshape(1, 2).contiguous()
    g = _f32c(wh2)
    out = torch.empty(1, g.shape[0], dtype=torch.float32, device=g.device)
    _lib.check(_lib.lib().odpl_bbox_wh_iou(_lib.ptr(a), _lib.ptr(g), _lib.ptr(out), 1, g.shape[0], _lib.current_stream(g.device)), "odpl_bbox_wh_iou")
    return out[0]


def box_iou_matrix(box1, box2, order="xyxy"):
    """RetinaNetLoss.box_iou (losses.py:375-403): [N,4] x [M,4] -> [N,M]."""
    _lib.require_cuda(box1, box2)
    b1, b2 = _f32c(box1), _f32c(box2)
    out = torch.empty(b1.shape[0], b2.shape[0], dtype=torch.float32, device=b1.device)
    _lib.check(_lib.lib().odpl_box_iou_matrix(_lib.ptr(b1), _lib.ptr(b2), _lib.ptr(out), b1.shape[0], b2.shape[0],
                                              1 if order == "xywh" else 0, _lib.current_stream(b1.device)), "odpl_box_iou_matrix")
    return out


def xywh2xyxy(x):
    """accuracy.py:289-295."""
    _lib.require_cuda(x)
    xi = _f32c(x).view(-1, 4)
    out = torch.empty_like(xi)
    _lib.check(_lib.lib().odpl_xywh2xyxy(_lib.ptr(xi), _lib.ptr(out), xi.shape[0], _lib.current_stream(xi.device)), "odpl_xywh2xyxy")
    return out.view(x.shape)
