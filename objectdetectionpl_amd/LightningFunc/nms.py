"""Host side of the batched NMS kernels (libodpl_hip.so).  The model classes' `non_max_suppression`
methods (model/YOLOV4.py:221-283, model/RetinaNet.py:117-178, ...) are thin wrappers over these."""
import ctypes

import torch

from .. import _lib

_ws_cache = {}


def _workspace(tag, dev, need):
    key = (tag, dev)
    ws = _ws_cache.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
        _ws_cache[key] = ws
        return ws, False
    return ws, True


def yolo_non_max_suppression(predictions, conf_thres=-0.0151, nms_thres=0.4, num_anchors=3, one_workgroup=False):
    """model/YOLOV4.py:221-283.  `predictions`: a raw head tensor or a list of them, [B, A*(5+C), G, G]
    (YOLOv5: [B, 3, G, G, 5+C], whose memory the reference reinterprets as [B, 3, 5+C, G, G]).
    The reference overwrites conf_thres with -0.0151 whatever the caller passes (YOLOV4.py:228).
    Returns List[Optional[Tensor[K,7]]] = (x1, y1, x2, y2, object_conf, class_score, class_pred).
    one_workgroup: A/B switch -- keep the one-workgroup-per-image kernel where the class-split pipeline is the default (>= 8 classes)."""
    if not isinstance(predictions, (list, tuple)):
        predictions = [predictions]
    _lib.require_cuda(*predictions)
    L = _lib.lib()
    dev = predictions[0].device
    d = _lib.YoloNmsDesc()
    heads = []
    for i, p in enumerate(predictions):
        p = p.float().contiguous()
        heads.append(p)
        if p.dim() == 5:            # v5 layout [B,3,G,G,no]: same memory, read as [B,3,no,G,G] (YOLOV5.py:176-182)
            A, G, no = p.shape[1], p.shape[2], p.shape[4]
        else:
            A = num_anchors
            G, no = p.shape[2], p.shape[1] // A
        d.head[i] = p.data_ptr()
        d.G[i] = G
    d.nscales, d.B, d.A, d.no = len(heads), heads[0].shape[0], A, no
    d.conf_thres, d.nms_thres = -0.0151, nms_thres
    need = L.odpl_yolo_nms_workspace_bytes(ctypes.byref(d))
    if need == 0:
        raise _lib.OdplError("odpl_yolo_nms_workspace_bytes rejected the descriptor")
    ws, _ = _workspace("yolo_nms", dev, need)
    N = sum(A * h.shape[2] * h.shape[2] if h.dim() == 4 else A * h.shape[2] * h.shape[3] for h in heads)
    out = torch.empty(d.B, N, 7, dtype=torch.float32, device=dev)
    counts = torch.empty(d.B, dtype=torch.int32, device=dev)
    d.flags = _lib.YOLO_NMS_ONE_WORKGROUP if one_workgroup else 0
    _lib.check(L.odpl_yolo_nms(ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.ptr(counts),
                               _lib.current_stream(dev)), "odpl_yolo_nms")
    ks = counts.tolist()          # the list-of-tensors API has data-dependent shapes: one sync, as in the reference
    return [None if k < 0 else out[b, :k] for b, k in enumerate(ks)]


def greedy_non_max_suppression(predictions, anchors, topk=100, nms_thresh=0.5, class_thresh=0.45, mode="union"):
    """model/RetinaNet.py:117-178 == model/SSD.py:249-310.  predictions = (loc_preds [B,An,4], cls_preds [B,An,C]).
    Returns List[Tensor[K,7]] = (x1, y1, x2, y2, 0, score, label)."""
    if mode != "union":
        raise TypeError("Unknown nms mode: %s." % mode) if mode != "min" else NotImplementedError("mode='min'")
    loc, cls = predictions
    _lib.require_cuda(loc, cls)
    L = _lib.lib()
    dev = loc.device
    loc, cls = loc.float().contiguous(), cls.float().contiguous()
    an = torch.as_tensor(anchors, dtype=torch.float32).to(dev).contiguous()
    d = _lib.GreedyNmsDesc()
    d.loc, d.cls, d.anchors = loc.data_ptr(), cls.data_ptr(), an.data_ptr()
    d.B, d.An, d.C, d.topk = loc.shape[0], loc.shape[1], cls.shape[2], topk
    d.nms_thresh, d.class_thresh = nms_thresh, class_thresh
    need = L.odpl_greedy_nms_workspace_bytes(ctypes.byref(d))
    if need == 0:
        raise _lib.OdplError("odpl_greedy_nms_workspace_bytes rejected the descriptor")
    ws, _ = _workspace("greedy_nms", dev, need)
    out = torch.empty(d.B, topk, 7, dtype=torch.float32, device=dev)
    counts = torch.empty(d.B, 2, dtype=torch.int32, device=dev)
    _lib.check(L.odpl_greedy_nms(ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.ptr(out), _lib.ptr(counts),
                                 _lib.current_stream(dev)), "odpl_greedy_nms")
    res = []
    for b, (k, n) in enumerate(counts.tolist()):
        if n == 1:      # the reference's `.nonzero().squeeze()` yields a 0-dim index there (RetinaNet.py:130-134)
            raise IndexError("too many indices for tensor of dimension 1")
        res.append(out[b, :k])
    return res
