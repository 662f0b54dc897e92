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


def get_batch_statistics(outputs, targets, iou_threshold):
    """accuracy.py:116-154.  outputs: List[Optional[Tensor[K,7]]] (NMS output), targets [nT,6] with columns 2:6
    already in pixels.  Returns the reference's list of [true_positives (np.float64), scores (np), labels (np)]
    for the images whose output is not None."""
    import numpy as np
    live = [(i, o) for i, o in enumerate(outputs) if o is not None]
    if not live:
        return []
    _lib.require_cuda(targets, *[o for _, o in live])
    dev = targets.device
    B = len(outputs)
    Kmax = max(1, max(o.shape[0] for _, o in live))
    dets = torch.zeros(B, Kmax, 7, dtype=torch.float32, device=dev)
    counts = torch.full((B,), -1, dtype=torch.int32)
    for i, o in live:
        dets[i, :o.shape[0]] = o
        counts[i] = o.shape[0]
    counts = counts.to(dev)
    tg = targets.float().contiguous()
    tp = torch.empty(B, Kmax, dtype=torch.float32, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    _lib.check(_lib.lib().odpl_batch_statistics(_lib.ptr(dets), _lib.ptr(counts), _lib.ptr(tg) if tg.shape[0] else None, B, Kmax,
                                                tg.shape[0], float(iou_threshold), _lib.ptr(tp), _lib.ptr(status),
                                                _lib.current_stream(dev)), "odpl_batch_statistics")
    tp_h = tp.cpu().numpy().astype(np.float64)
    return [[tp_h[i, :o.shape[0]], o[:, 4].cpu().data.numpy(), o[:, -1].cpu().data.numpy()] for i, o in live]


def ap_per_class(tp, conf, pred_cls, target_cls, device="cuda"):
    """accuracy.py:207-260 (+ compute_ap 262-287): numpy in, numpy out (p, r, ap, f1, unique_classes int32), computed
    on the device in double precision."""
    import ctypes as _ct

    import numpy as np
    dev = torch.device(device)
    N, M = len(tp), len(target_cls)
    if M == 0:
        e = np.array([])
        return e, e, e, e, np.array([], dtype="int32")
    if N == 0:
        uc = np.unique(np.asarray(target_cls))
        z = np.zeros(len(uc))
        return z, z.copy(), z.copy(), z.copy(), uc.astype("int32")
    L = _lib.lib()
    t_tp = torch.as_tensor(np.asarray(tp, dtype=np.float64)).to(dev)
    t_cf = torch.as_tensor(np.asarray(conf, dtype=np.float32)).to(dev)
    t_pc = torch.as_tensor(np.asarray(pred_cls, dtype=np.float32)).to(dev)
    t_tc = torch.as_tensor(np.asarray(target_cls, dtype=np.float32)).to(dev)
    need = L.odpl_ap_per_class_workspace_bytes(N, M)
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    outs = [torch.zeros(M, dtype=torch.float64, device=dev) for _ in range(4)]
    ocls = torch.zeros(M, dtype=torch.int32, device=dev)
    ncls = torch.zeros(1, dtype=torch.int32, device=dev)
    _lib.check(L.odpl_ap_per_class(_lib.ptr(t_tp), _lib.ptr(t_cf), _lib.ptr(t_pc), _lib.ptr(t_tc), N, M, _lib.ptr(ws), need,
                                   *[_lib.ptr(o) for o in outs], _lib.ptr(ocls), _lib.ptr(ncls), _lib.current_stream(dev)),
               "odpl_ap_per_class")
    k = int(ncls.item())
    p, r, ap, f1 = (o[:k].cpu().numpy() for o in outs)
    return p, r, ap, f1, ocls[:k].cpu().numpy().astype("int32")


def get_yolo_statistics(self, output, target, output_to_cpu=True):
    """accuracy.py:382-470 (attached to the YOLO v2/v3/v4 model classes as a method, `self` = the model).
    Returns {grid_size: [cls_acc, recall50, recall75, precision, conf_obj, conf_noobj, output.cpu()]}.
    `output_to_cpu=False` leaves the 7th element (the decoded [B, A*G*G, 5+C] tensor, 58 MB per bs-16 batch, which
    `test_epoch_end` drops unread -- step.py:139) on the device: the D2H copy is 7 ms of a 20 ms test step."""
    import ctypes as _ct
    if type(output) != list:
        output = [output]
    L = _lib.lib()
    res = {}
    for i, x in enumerate(output):
        if self.anch_masks is not None:                                  # yolo_v4
            anchors = [self.anchors[m] for m in self.anch_masks[i]]
        elif len(self.anchors) == 3:                                     # yolo_v3
            anchors = self.anchors[i]
        else:                                                            # yolo_v2
            anchors = self.anchors
        _lib.require_cuda(x, target)
        x = x.float().contiguous()
        dev = x.device
        A, B, G, C = len(anchors), x.size(0), x.size(2), self.num_classes
        stride = self.img_size / G
        wh = (_ct.c_float * (2 * A))()
        for a, (aw, ah) in enumerate(anchors):
            wh[2 * a], wh[2 * a + 1] = aw / stride, ah / stride            # accuracy.py:426: python double, then f32
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        pb, pconf, pcls, out = f(B, A, G, G, 4), f(B, A, G, G), f(B, A, G, G, C), f(B, A * G * G, 5 + C)
        _lib.check(L.odpl_yolo_train_decode(_lib.ptr(x), B, A, C, G, wh, stride, _lib.ptr(pb), _lib.ptr(pconf), _lib.ptr(pcls),
                                            _lib.ptr(out), _lib.current_stream(dev)), "odpl_yolo_train_decode")
        sa = torch.tensor([[wh[2 * a], wh[2 * a + 1]] for a in range(A)], dtype=torch.float32, device=dev)
        iou_scores, class_mask, obj_mask, noobj_mask, *_rest = build_targets(pb, pcls, target, sa, self.ignore_thres)
        out6 = f(6)
        _lib.check(L.odpl_yolo_stats_reduce(_lib.ptr(pconf), _lib.ptr(class_mask), _lib.ptr(iou_scores), _lib.ptr(obj_mask),
                                            _lib.ptr(noobj_mask), B * A * G * G, _lib.ptr(out6), _lib.current_stream(dev)),
                   "odpl_yolo_stats_reduce")
        vals = out6.cpu().numpy()
        res[G] = [vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], out.cpu() if output_to_cpu else out]
    return res
