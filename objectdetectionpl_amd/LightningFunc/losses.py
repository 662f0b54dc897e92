"""Host-side mirror of the reference's `LightningFunc/losses.py` (same names, ctor signatures,
`forward(output, target[, seen]) -> dict` with the same keys); every forward is one C-ABI call
into libodpl_hip.so (hand-written gfx950 kernels).  No eager-torch arithmetic, no CPU fallback.

    configure_loss                 losses.py:9-35
    RegionLoss_v4 / MultiScaleRegionLoss_v4      losses.py:475-634
    RegionLoss_v3 / MultiScaleRegionLoss_v3      losses.py:636-795
    RegionLoss_v2                                losses.py:797-905
"""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .._lib import OdplError
from .utils.headconv import fusable_source, structured_backward_supported

KEYS = ("loss", "Localization", "Size", "Conf", "Classification", "Conf_obj", "Conf_noobj")


def _criterion_name(c, table):
    """Accept the reference's criterion CLASSES (nn.BCELoss, nn.MSELoss, ...) or their config names."""
    if isinstance(c, str):
        name = c
    else:
        name = {nn.BCELoss: "bce_loss", nn.CrossEntropyLoss: "ce_loss", nn.MSELoss: "mse_loss",
                nn.SmoothL1Loss: "smooth_l1_loss"}.get(c)
        if name is None and getattr(c, "__name__", "") == "focal_loss":
            # the reference instantiates cls_criterion() with no arguments (losses.py:485), which for
            # focal_loss(num_classes, ...) is a TypeError at construction
            raise TypeError("__init__() missing 1 required positional argument: 'num_classes'")
    if name not in table:
        raise ValueError(f"unsupported criterion {c!r} for the YOLO region loss")
    return name


def _anchor_table(anchors, img_dim, grid):
    """losses.py:499-503: Python-double stride and division, rounded to fp32 once."""
    stride = img_dim / grid
    return [(ctypes.c_float(aw / stride).value, ctypes.c_float(ah / stride).value) for aw, ah in anchors]


def _memory_format(h):
    """-> (tensor to hand to the library, element strides (b, c, y, x)).  NCHW-contiguous and channels-last tensors are
    consumed in place (the producing convolution's own format); anything else is copied to NCHW once."""
    if h.is_contiguous():
        return h, (0, 0, 0, 0)
    if h.dim() == 4 and h.is_contiguous(memory_format=torch.channels_last):
        return h, tuple(h.stride())
    return h.contiguous(), (0, 0, 0, 0)


class _RegionLossFn(torch.autograd.Function):
    """loss (differentiable) + the six metric scalars.  forward leaves d loss / d head in the library's compact form
    (one value per cell + 4+C values per obj cell); backward expands it into dense tensors in the heads' own memory
    format, scaled by the upstream gradient on the device -- one launch, no host sync."""

    @staticmethod
    def forward(ctx, owner, target, scale_anchors, average, *heads):
        need_grad = any(ctx.needs_input_grad[4:])
        pairs = [_memory_format(h) for h in heads]
        d, keep = owner._describe([p[0] for p in pairs], [p[1] for p in pairs], target, scale_anchors, average)
        out7, counts, compact = owner._forward(d, pairs[0][0].device, need_grad)
        ctx.desc, ctx.keep, ctx.compact = d, keep, compact
        ctx.like = [p[0] for p in pairs] if need_grad else None
        ctx.used = False
        loss = out7[0]
        metrics = out7[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return loss, metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _g_metrics, _g_counts):
        if ctx.used:
            raise RuntimeError("the fused region loss frees its gradient buffers after the first backward "
                               "(call backward once per forward)")
        ctx.used = True
        d, compact, like = ctx.desc, ctx.compact, ctx.like
        ctx.desc = ctx.compact = ctx.like = ctx.keep = None
        grads = [torch.empty_like(h) if need else None for h, need in zip(like, ctx.needs_input_grad[4:])]    # preserves the strides
        for i, g in enumerate(grads):
            d.scale[i].grad = g.data_ptr() if g is not None else None
        up = g_loss.contiguous().float()
        dev = like[0].device
        _lib.check(_lib.lib().odpl_yolo_region_loss_backward(ctypes.byref(d), _lib.ptr(compact), compact.numel(), _lib.ptr(up),
                                                             _lib.current_stream(dev)), "odpl_yolo_region_loss_backward")
        return (None, None, None, None) + tuple(grads)


class _RegionLossHeadFn(torch.autograd.Function):
    """The same loss when every head is the output of `utils.headconv.head_conv`: differentiable inputs are the head
    convolutions' (x, weight, bias), not the head tensors.  backward runs each convolution's backward from the compact
    gradient (`odpl_head_conv_backward`): the dense d loss/d head tensors are never materialised."""

    @staticmethod
    def forward(ctx, owner, target, scale_anchors, average, n, *args):
        heads, srcs = args[:n], args[n:]
        d, keep = owner._describe(list(heads), [(0, 0, 0, 0)] * n, target, scale_anchors, average)
        out7, counts, compact = owner._forward(d, heads[0].device, True)
        ctx.desc, ctx.keep, ctx.compact, ctx.n = d, keep, compact, n
        # (x, w, b) per head through save_for_backward: autograd's version counters then refuse a backward after an in-place
        # edit of an activation or a parameter (a bias may be absent)
        ctx.has = [t is not None for t in srcs]
        ctx.save_for_backward(*[t for t in srcs if t is not None])
        ctx.used = False
        loss = out7[0]
        metrics = out7[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return loss, metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _g_metrics, _g_counts):
        from .utils.headconv import head_conv_backward
        if ctx.used:
            raise RuntimeError("the fused region loss frees its gradient buffers after the first backward "
                               "(call backward once per forward)")
        ctx.used = True
        d, compact, n = ctx.desc, ctx.compact, ctx.n
        saved = iter(ctx.saved_tensors)
        srcs = [next(saved) if h else None for h in ctx.has]
        xs, ws, bs = srcs[0::3], srcs[1::3], srcs[2::3]
        ctx.desc = ctx.compact = ctx.keep = None
        up = g_loss.contiguous().float()
        L = _lib.lib()
        need_all = ctx.needs_input_grad[5 + n:]
        sparses = []
        for i in range(n):
            sg = _lib.SparseGrad()
            _lib.check(L.odpl_yolo_compact_view(ctypes.byref(d), _lib.ptr(compact), i, ctypes.byref(sg)), "odpl_yolo_compact_view")
            sparses.append(sg)
        out = head_conv_backward(sparses, xs, ws, bs, up, [need_all[3 * i:3 * i + 3] for i in range(n)])
        return (None, None, None, None, None) + (None,) * n + tuple(out)


class _RegionLossBase(nn.Module):
    """Shared host logic of the v2/v3/v4 region losses."""

    def __init__(self, anchors, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim=416):
        super().__init__()
        self.anchors = anchors
        self.num_anchors = len(anchors)
        self.num_classes = num_classes
        self.ignore_thres = 0.5          # losses.py:483
        self.obj_scale = 1               # losses.py:487
        self.noobj_scale = 100           # losses.py:488
        self.metrics = {}
        self.img_dim = img_dim
        self.grid_size = 0
        self._cls_name = _criterion_name(cls_criterion, _lib.CLS_MODES)
        self._coord_name = _criterion_name(coord_criterion, _lib.COORD_MODES)
        self._ws = None
        self._ws_clean_key = None        # (device, stream, shapes, formats) for which the workspace's state maps are known clean
        self.last_counts = None          # int32 [nscales, 8] on device: nObj, nNoobj, guard1, guard2, status

    # class switch: False = always hand autograd the dense d loss/d head (the head convolutions then run their own backward)
    fuse_head_backward = True

    # -- descriptor ---------------------------------------------------------------------------------
    def _describe(self, heads, strides, target, scale_anchors, average):
        _lib.require_cuda(target, *heads)
        C = self.num_classes
        d = _lib.YoloLossDesc()
        d.B = heads[0].shape[0]
        d.C = C
        tgt = target
        if tgt.dtype != torch.float32 or not tgt.is_contiguous():
            tgt = tgt.float().contiguous()
        if tgt.dim() != 2 or (tgt.shape[0] and tgt.shape[1] != 6):
            raise ValueError(f"target must be [nT,6] (img, cls, x, y, w, h); got {tuple(target.shape)}")
        d.nT = tgt.shape[0]
        d.nscales = len(heads)
        if d.nscales > _lib.MAX_SCALES:
            raise ValueError(f"at most {_lib.MAX_SCALES} scales")
        d.coord_mode = _lib.COORD_MODES[self._coord_name]
        d.cls_mode = _lib.CLS_MODES[self._cls_name]
        d.average = 1 if average else 0
        d.ignore_thres = self.ignore_thres
        d.obj_scale = self.obj_scale
        d.noobj_scale = self.noobj_scale
        d.target = tgt.data_ptr() if d.nT else None
        for i, (h, st, anc) in enumerate(zip(heads, strides, scale_anchors)):
            A = len(anc)
            if h.dtype != torch.float32:
                raise TypeError("head tensors must be fp32 (the reference computes the loss in fp32)")
            if h.dim() != 4 or h.shape[1] != A * (5 + C) or h.shape[2] != h.shape[3] or h.shape[0] != d.B:
                raise ValueError(f"head {i}: expected [B,{A * (5 + C)},G,G], got {tuple(h.shape)}")
            if A > _lib.MAX_ANCHORS:
                raise ValueError(f"at most {_lib.MAX_ANCHORS} anchors per scale")
            s = d.scale[i]
            s.head = h.data_ptr()
            s.grad = None
            s.G = h.shape[2]
            s.A = A
            for k in range(4):
                s.stride[k] = st[k]
            for a, (aw, ah) in enumerate(_anchor_table(anc, self.img_dim, s.G)):
                s.anchor_w[a] = aw
                s.anchor_h[a] = ah
        return d, (tgt, heads)

    # -- forward: one C-ABI call ----------------------------------------------------------------------
    def _forward(self, d, dev, need_grad):
        L = _lib.lib()
        need = L.odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d))          # every grad is NULL: workspace without the compact area
        if need == 0:
            raise OdplError("odpl_yolo_region_loss_workspace_bytes rejected the descriptor")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
            self._ws_clean_key = None
        key = (dev, torch.cuda.current_stream(dev).cuda_stream, d.B,
               tuple((s.G, s.A, tuple(s.stride)) for s in d.scale[:d.nscales]))
        # under stream capture the kernels are only RECORDED: the workspace is not cleaned by this call, and the graph must
        # carry its own fill so that every replay starts from a clean map -- never claim STATE_CLEAN there, and leave the
        # eager bookkeeping as it was
        capturing = torch.cuda.is_current_stream_capturing()
        d.flags = _lib.YOLO_STATE_CLEAN if (self._ws_clean_key == key and not capturing) else 0
        if not capturing:
            self._ws_clean_key = None    # stays None if the call below raises
        compact = None
        if need_grad:
            compact = torch.empty(L.odpl_yolo_region_loss_compact_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev)
        out7 = torch.empty(7, dtype=torch.float32, device=dev)
        counts = torch.empty(d.nscales, 8, dtype=torch.int32, device=dev)
        _lib.check(L.odpl_yolo_region_loss_forward(ctypes.byref(d), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(compact),
                                                   0 if compact is None else compact.numel(), _lib.ptr(out7), None, _lib.ptr(counts),
                                                   _lib.current_stream(dev)), "odpl_yolo_region_loss_forward")
        if not capturing:
            self._ws_clean_key = key     # the conf blocks reset every cell they read
        self.last_counts = counts
        return out7, counts, compact

    def _run(self, heads, target, scale_anchors, average):
        # half-precision heads (a bf16 / fp16 autocast network): the reference computes this loss in fp32, so widen once;
        # autograd narrows the gradients back (YOLOv5's criterion consumes bf16 heads natively instead)
        srcs = [fusable_source(h) for h in heads] if self.fuse_head_backward else [None]
        fuse = all(s is not None for s in srcs) and all(h.dtype == torch.float32 and h.is_contiguous() for h in heads)
        if fuse:        # the structured head backward has limits of its own (anchors per scale, tile geometry): ask it now
            nT = target.shape[0] if target.dim() == 2 else -1
            geoms = [(h.shape[0], h.shape[2] * h.shape[3], len(anc), 5 + self.num_classes, 4, nT, _lib.LAYOUT_PLANAR)
                     for h, anc in zip(heads, scale_anchors)]
            fuse = nT >= 0 and structured_backward_supported(srcs, geoms)
        if fuse:
            flat = [t for s in srcs for t in s]
            loss, metrics, _counts = _RegionLossHeadFn.apply(self, target, scale_anchors, average, len(heads),
                                                             *[h.detach() for h in heads], *flat)
        else:
            heads = [h.float() if h.dtype in (torch.bfloat16, torch.float16) else h for h in heads]
            loss, metrics, _counts = _RegionLossFn.apply(self, target, scale_anchors, average, *heads)
        vals = (loss,) + tuple(metrics.unbind(0))
        return dict(zip(KEYS, vals))


class RegionLoss_v4(_RegionLossBase):
    """Single-scale YOLOv4 region loss, losses.py:475-575.  forward returns the reference's 7-tuple."""

    def forward(self, x, targets=None):
        self.grid_size = x.size(2)
        anc = [tuple(a) for a in self.anchors]          # 3 per scale: the reference views the head with 3 (:519)
        r = self._run([x], targets, [anc], average=False)
        return tuple(r[k] for k in KEYS)


class MultiScaleRegionLoss_v4(RegionLoss_v4):
    """losses.py:577-634: anchors picked per scale through anch_masks, entries averaged over the scales."""

    def __init__(self, anchors, anch_masks, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim=416, **kwargs):
        super().__init__(anchors, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim, **kwargs)
        self._anchors = anchors
        self._anch_masks = anch_masks

    def forward(self, output, target, seen=None):
        scale_anchors = [[tuple(self._anchors[m]) for m in self._anch_masks[i]] for i in range(len(output))]
        self.anchors = scale_anchors[-1]
        self.num_anchors = len(self.anchors)
        r = self._run(list(output), target, scale_anchors, average=True)
        self.loss_tot, self.loss_coord, self.loss_size = r["loss"], r["Localization"], r["Size"]
        self.loss_conf, self.loss_cls = r["Conf"], r["Classification"]
        self.loss_conf_obj, self.loss_conf_noobj = r["Conf_obj"], r["Conf_noobj"]
        return r


class RegionLoss_v3(_RegionLossBase):
    """losses.py:636-736 (identical arithmetic to v4)."""

    def forward(self, x, targets=None):
        self.grid_size = x.size(2)
        anc = [tuple(a) for a in self.anchors]
        r = self._run([x], targets, [anc], average=False)
        return tuple(r[k] for k in KEYS)


class MultiScaleRegionLoss_v3(RegionLoss_v3):
    """losses.py:738-795.  `anchors` is the per-scale list the YOLOv3 ctor already divided by the stride
    (YOLOV3.py:54-56); compute_grid_offsets divides by the stride again (losses.py:661-665) -- kept."""

    def __init__(self, anchors, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim=416, **kwargs):
        super().__init__(anchors, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim, **kwargs)
        self._anchors = anchors

    def forward(self, output, target, seen=None):
        scale_anchors = [[tuple(a) for a in self._anchors[i]] for i in range(len(output))]
        self.anchors = self._anchors[len(output) - 1]
        self.num_anchors = len(self.anchors)
        r = self._run(list(output), target, scale_anchors, average=True)
        self.loss_tot, self.loss_coord, self.loss_size = r["loss"], r["Localization"], r["Size"]
        self.loss_conf, self.loss_cls = r["Conf"], r["Classification"]
        self.loss_conf_obj, self.loss_conf_noobj = r["Conf_obj"], r["Conf_noobj"]
        return r


class RegionLoss_v2(_RegionLossBase):
    """losses.py:797-905: one scale, A = len(anchors) (5 for YOLOv2), dict returned directly."""

    def forward(self, x, targets=None):
        self.grid_size = x.size(2)
        anc = [tuple(a) for a in self.anchors]
        return self._run([x], targets, [anc], average=False)


def configure_loss(args, iou_boxes, anchors, anch_masks, num_classes, img_size):
    """losses.py:9-35.  Same dispatch on args.model_name / args.cls_criterion / args.coord_criterion."""
    cls_name = args.cls_criterion
    coord_name = args.coord_criterion
    if cls_name not in ("focal_loss", "ce_loss", "bce_loss"):
        raise UnboundLocalError("local variable 'cls_criterion' referenced before assignment")
    if coord_name not in ("mse_loss", "smooth_l1_loss"):
        raise UnboundLocalError("local variable 'coord_criterion' referenced before assignment")
    conf = "bce_loss"
    name = args.model_name
    if name in ("RetinaNet", "SSD", "YOLOv5"):
        from . import losses_dense
        return losses_dense.configure(args, iou_boxes, anchors, anch_masks, num_classes, img_size)
    if cls_name == "focal_loss":
        _criterion_name(type("focal_loss", (), {}), _lib.CLS_MODES)     # raises the reference's TypeError
    if name == "YOLOv4":
        return MultiScaleRegionLoss_v4(anchors, anch_masks, cls_name, coord_name, conf, num_classes, img_size)
    if name == "YOLOv3":
        return MultiScaleRegionLoss_v3(anchors, cls_name, coord_name, conf, num_classes, img_size)
    if name == "YOLOv2":
        return RegionLoss_v2(anchors, cls_name, coord_name, conf, num_classes, img_size)
    return None   # the reference falls off the end of its if-chain for unknown names
