"""Host-side mirrors of the anchor-dense criteria of `LightningFunc/losses.py`: RetinaNetLoss
(losses.py:351-472), focal_loss (317-349), SSDLoss (159-315), MultiScaleRegionLoss_v5 (70-156).
Each forward is one C-ABI call into libodpl_hip.so; there is no CPU fallback."""
import ctypes

import torch
import torch.nn as nn

from .. import _lib
from .._lib import OdplError


class focal_loss(nn.Module):
    """Marker + standalone module for the reference's SOFTMAX focal loss (losses.py:317-349).
    `configure_loss` passes this CLASS as cls_criterion, exactly like the reference."""

    def __init__(self, num_classes, reduction="sum"):
        super().__init__()
        self.num_classes = num_classes
        self.reduction = reduction


def _coord_name(c):
    if isinstance(c, str):
        return c
    return {nn.MSELoss: "mse_loss", nn.SmoothL1Loss: "smooth_l1_loss"}[c]


class _RetinaLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, target, loc, cls):
        need = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        loc_c, cls_c = loc.contiguous(), cls.contiguous()
        g_loc = torch.empty_like(loc_c) if need else None
        g_cls = torch.empty_like(cls_c) if need else None
        out3, counts = owner._launch(loc_c, cls_c, g_loc, g_cls, target)
        ctx.grads = (g_loc, g_cls)
        ctx.used = False
        metrics = out3[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out3[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        if ctx.used:
            raise RuntimeError("the fused RetinaNet loss frees its gradient buffers after the first backward")
        ctx.used = True
        g_loc, g_cls = ctx.grads
        ctx.grads = None
        g = g_loss.contiguous().float()
        L = _lib.lib()
        for gr in (g_loc, g_cls):
            if gr is not None:
                _lib.check(L.odpl_scale_inplace_if_not_one(_lib.ptr(gr), gr.numel(), _lib.ptr(g),
                                                           _lib.current_stream(gr.device)), "odpl_scale_inplace_if_not_one")
        return None, None, g_loc, g_cls


class _RetinaLossHeadFn(torch.autograd.Function):
    """The same loss when (loc_preds, cls_preds) come from `utils.retinahead.final_convs`: the differentiable inputs are the two final
    head convolutions' weights, biases and per-level inputs.  forward writes d loss/d preds on the positive rows only (no zero
    stream); backward runs both convolutions' backward from those rows (`odpl_retina_heads_backward`)."""

    @staticmethod
    def forward(ctx, owner, target, loc, cls, src, held, *diff):
        B, An = loc.shape[0], loc.shape[1]
        dev = loc.device
        g_loc, g_cls = torch.empty_like(loc), torch.empty_like(cls)          # only the positive rows will be written -- and read
        if held is not None:                                                  # the assignment was made before the heads ran (owner.assign)
            pos, npos = held[2], held[3]
            out3, counts = owner._launch(loc, cls, g_loc, g_cls, target, rows_only=(pos, npos), mode="assigned")
        else:
            pos = torch.empty(B * An, dtype=torch.int32, device=dev)
            npos = torch.empty(1, dtype=torch.int32, device=dev)
            out3, counts = owner._launch(loc, cls, g_loc, g_cls, target, rows_only=(pos, npos))
        ctx.src, ctx.aux = src, (g_loc, g_cls, pos, npos)
        ctx.has = [t is not None for t in diff]
        ctx.save_for_backward(*[t for t in diff if t is not None])              # version-checked by autograd
        ctx.used = False
        metrics = out3[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out3[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        from .utils.retinahead import heads_backward
        if ctx.used:
            raise RuntimeError("the fused RetinaNet loss frees its gradient buffers after the first backward")
        ctx.used = True
        g_loc, g_cls, pos, npos = ctx.aux
        src = ctx.src
        ctx.aux = ctx.src = None
        _ = ctx.saved_tensors                                                    # raises if an input was edited in place since forward
        need = ctx.needs_input_grad[6:]
        gw, gx = heads_backward(src, g_loc, g_cls, pos, npos, g_loss.contiguous().float(), need[:4], need[4:])
        return (None, None, None, None, None, None) + tuple(gw) + tuple(gx)


class RetinaNetLoss(nn.Module):
    """losses.py:351-472.  ctor `(iou_boxes, cls_criterion, coord_criterion, num_classes, img_size)`;
    forward((loc_preds, cls_preds), targets) -> {"loss", "Localization", "Classification"}.
    An image without any box makes the reference raise (max over an empty dim): here the three scalars
    come back NaN and `last_counts[1] & 1` is set (no host sync on the hot path).
    Extension: `img_size=(width, height)` scales the x / w target columns by width and the y / h columns by height (the
    reference multiplies all four by one scalar, losses.py:425, which is only right for square inputs; BASELINE config 4 is
    1333 x 800).  A scalar -- the default -- is the reference's behaviour bit for bit."""

    def __init__(self, iou_boxes, cls_criterion, coord_criterion, num_classes=20, img_size=600):
        super().__init__()
        self.num_classes = num_classes
        self.img_size = img_size
        self.iou_boxes = iou_boxes
        if not (cls_criterion is focal_loss or getattr(cls_criterion, "__name__", cls_criterion) == "focal_loss"):
            # the reference calls cls_criterion(num_classes, reduction='sum'): only focal_loss has that signature
            raise TypeError("RetinaNetLoss needs cls_criterion=focal_loss (nn.BCELoss/nn.CrossEntropyLoss reject "
                            "(num_classes, reduction=...) positional arguments in the reference too)")
        self._coord_name = _coord_name(coord_criterion)
        self._ws = None
        self._clean_key = None
        self._anchors_dev = None
        self.last_counts = None

    def _anchors_on(self, dev):
        a = self._anchors_dev
        if a is None or a.device != dev:
            a = torch.as_tensor(self.iou_boxes, dtype=torch.float32).to(dev).contiguous()
            self._anchors_dev = a
        return a

    fuse_head_backward = True      # class switch: False = always hand autograd the dense d loss/d preds

    def assign(self, target, B):
        """Target assignment alone (losses.py:423-446), BEFORE any prediction exists: -> (pos_anchor, num_pos, token).  The next
        forward whose predictions carry this token (utils/retinahead.final_convs(..., assignment=token)) reuses the assignment held
        in the workspace instead of running it again."""
        _lib.require_cuda(target)
        dev = target.device
        An = self._anchors_on(dev).shape[0]
        pos = torch.empty(B * An, dtype=torch.int32, device=dev)
        npos = torch.empty(1, dtype=torch.int32, device=dev)
        tgt = target.float().contiguous()
        self._launch(None, None, None, None, tgt, rows_only=(pos, npos), mode="assign", B=B)
        token = object()
        self._assignment = (token, tgt, pos, npos, target)
        return pos, npos, token

    def _launch(self, loc, cls, g_loc, g_cls, target, rows_only=None, mode=None, B=None):
        L = _lib.lib()
        if mode == "assign":
            dev = target.device
            anchors = self._anchors_on(dev)
            An, C = anchors.shape[0], self.num_classes
        else:
            _lib.require_cuda(loc, cls, target)
            dev = loc.device
            if loc.dtype != torch.float32 or cls.dtype != torch.float32:
                raise TypeError("loc/cls predictions must be fp32")
            anchors = self._anchors_on(dev)
            B, An = loc.shape[0], loc.shape[1]
            C = self.num_classes
            if loc.shape != (B, An, 4) or cls.shape != (B, An, C) or anchors.shape != (An, 4):
                raise ValueError(f"shape mismatch: loc {tuple(loc.shape)}, cls {tuple(cls.shape)}, anchors {tuple(anchors.shape)}")
        tgt = target.float().contiguous()
        d = _lib.RetinaLossDesc()
        d.loc, d.cls = (loc.data_ptr(), cls.data_ptr()) if loc is not None else (None, None)
        d.grad_loc = g_loc.data_ptr() if g_loc is not None else None
        d.grad_cls = g_cls.data_ptr() if g_cls is not None else None
        d.anchors = anchors.data_ptr()
        d.target = tgt.data_ptr() if tgt.shape[0] else None
        d.B, d.An, d.C, d.nT = B, An, C, tgt.shape[0]
        d.coord_mode = _lib.COORD_MODES[self._coord_name]
        if isinstance(self.img_size, (tuple, list)):        # extension: (width, height) of a non-square input; the reference has one scalar
            d.img_size, d.img_size_y = float(self.img_size[0]), float(self.img_size[1])
        else:
            d.img_size, d.img_size_y = float(self.img_size), 0.0
        if mode == "assign":
            d.flags = _lib.RETINA_ASSIGN_ONLY
            d.pos_anchor, d.num_pos = rows_only[0].data_ptr(), rows_only[1].data_ptr()
        need = L.odpl_retina_loss_workspace_bytes(ctypes.byref(d))
        if need == 0:
            raise OdplError("odpl_retina_loss_workspace_bytes rejected the descriptor")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
            self._clean_key = None
        key = (dev, torch.cuda.current_stream(dev).cuda_stream)
        d.flags = _lib.RETINA_STATE_CLEAN if self._clean_key == key else 0
        if rows_only is not None:              # gradient on the positive rows only + the list of those rows (structured backward)
            d.flags |= _lib.RETINA_ROWS_ONLY
            d.pos_anchor, d.num_pos = rows_only[0].data_ptr(), rows_only[1].data_ptr()
        self._clean_key = None
        if mode == "assign":
            d.flags |= _lib.RETINA_ASSIGN_ONLY
            _lib.check(L.odpl_retina_loss(ctypes.byref(d), _lib.ptr(self._ws), self._ws.numel(), None, None, _lib.current_stream(dev)),
                       "odpl_retina_loss")
            self._assigned_ws = (self._ws.data_ptr(), key)
            return None, None
        if mode == "assigned":
            if getattr(self, "_assigned_ws", None) != (self._ws.data_ptr(), key):
                raise OdplError("the assignment this forward was promised is gone (workspace or stream changed since RetinaNetLoss.assign)")
            d.flags = (d.flags & ~_lib.RETINA_STATE_CLEAN) | _lib.RETINA_ASSIGNED
        self._assigned_ws = None
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        counts = torch.empty(4, dtype=torch.int32, device=dev)
        _lib.check(L.odpl_retina_loss(ctypes.byref(d), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(out3),
                                      _lib.ptr(counts), _lib.current_stream(dev)), "odpl_retina_loss")
        self._clean_key = key
        self.last_counts = counts
        return out3, counts

    def forward(self, outputs, targets):
        from .utils.retinahead import fusable_source
        loc_preds, cls_preds = outputs
        src = fusable_source(loc_preds, cls_preds) if self.fuse_head_backward else None
        promised = getattr(loc_preds, "_odpl_retina_src", None)
        if promised is not None and promised.get("assignment") is not None:
            # predictions of a loss-driven step: they exist on this assignment's positive rows only -- no other consumer is valid
            held = getattr(self, "_assignment", None)
            self._assignment = None
            if src is None or held is None or held[0] is not promised["assignment"] or held[4] is not targets:
                raise OdplError("these predictions were computed for one RetinaNetLoss.assign call only (its positive rows, its targets); "
                                "run the dense forward for any other consumer")
            loss, metrics, _ = _RetinaLossHeadFn.apply(self, held[1], loc_preds.detach(), cls_preds.detach(), src, held, *src["w"], *src["xs"])
            return {"loss": loss, "Localization": metrics[0], "Classification": metrics[1]}
        if src is not None:
            loss, metrics, _ = _RetinaLossHeadFn.apply(self, targets, loc_preds.detach(), cls_preds.detach(), src, None, *src["w"], *src["xs"])
            return {"loss": loss, "Localization": metrics[0], "Classification": metrics[1]}
        loc_preds, cls_preds = (_widen(t) for t in outputs)
        loss, metrics, _ = _RetinaLossFn.apply(self, targets, loc_preds, cls_preds)
        return {"loss": loss, "Localization": metrics[0], "Classification": metrics[1]}


def _widen(t):
    """Half-precision predictions (an autocast network): the reference computes these losses in fp32 -- widen once, autograd
    narrows the gradient back."""
    return t.float() if t.dtype in (torch.bfloat16, torch.float16) else t


class _SSDLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, owner, target, off, cls):
        need = ctx.needs_input_grad[2] or ctx.needs_input_grad[3]
        off_c, cls_c = off.contiguous(), cls.contiguous()
        g_off = torch.empty_like(off_c) if need else None
        g_cls = torch.empty_like(cls_c) if need else None
        out3, counts = owner._launch(off_c, cls_c, g_off, g_cls, target)
        ctx.grads = (g_off, g_cls)
        ctx.used = False
        metrics = out3[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out3[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        if ctx.used:
            raise RuntimeError("the fused SSD loss frees its gradient buffers after the first backward")
        ctx.used = True
        g_off, g_cls = ctx.grads
        ctx.grads = None
        g = g_loss.contiguous().float()
        L = _lib.lib()
        for gr in (g_off, g_cls):
            if gr is not None:
                _lib.check(L.odpl_scale_inplace_if_not_one(_lib.ptr(gr), gr.numel(), _lib.ptr(g),
                                                           _lib.current_stream(gr.device)), "odpl_scale_inplace_if_not_one")
        return None, None, g_off, g_cls


class _SSDLossHeadFn(torch.autograd.Function):
    """The same loss when (out_bbx, out_cl) come from `utils.ssdhead.head_pairs`: the differentiable inputs are the six sources and the
    twelve head convolutions' weights and biases.  forward writes d loss/d preds on the positives and the mined negatives only (no zero
    stream over 8 732 x (C + 4) per image) and lists those priors; backward cuts the list by source and runs both convolutions'
    backward of every source from those rows (`odpl_retina_heads_backward`)."""

    @staticmethod
    def forward(ctx, owner, target, off, cls, src, *diff):
        B, Pn = off.shape[0], off.shape[1]
        dev = off.device
        g_off, g_cls = torch.empty_like(off), torch.empty_like(cls)            # only the active rows will be written -- and read
        active = torch.empty(B * Pn, dtype=torch.int32, device=dev)
        active_count = torch.empty(B, dtype=torch.int32, device=dev)
        out3, counts = owner._launch(off, cls, g_off, g_cls, target, rows_only=(active, active_count))
        ctx.src, ctx.aux = src, (g_off, g_cls, active, active_count)
        ctx.save_for_backward(*[t for t in diff if t is not None])              # version-checked by autograd
        ctx.used = False
        metrics = out3[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out3[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        from .utils import ssdhead
        if ctx.used:
            raise RuntimeError("the fused SSD loss frees its gradient buffers after the first backward")
        ctx.used = True
        g_off, g_cls, active, active_count = ctx.aux
        src = ctx.src
        ctx.aux = ctx.src = None
        _ = ctx.saved_tensors                                                    # raises if an input was edited in place since forward
        n = src["n"]
        need = ctx.needs_input_grad[5:]
        dev = g_off.device
        B, Pn = g_off.shape[0], g_off.shape[1]
        first = ssdhead.level_first(src)
        lists = torch.empty(B * Pn, dtype=torch.int32, device=dev)
        lcounts = torch.empty(n, dtype=torch.int32, device=dev)
        arr = (ctypes.c_int32 * (n + 1))(*first)
        _lib.check(_lib.lib().odpl_ssd_level_lists(_lib.ptr(active), _lib.ptr(active_count), B, Pn, n, arr, _lib.ptr(lists), _lib.ptr(lcounts),
                                                   _lib.current_stream(dev)), "odpl_ssd_level_lists")
        gx, gw = ssdhead.heads_backward(src, g_off, g_cls, lists, lcounts, g_loss.contiguous().float(), need[:n], need[n:])
        return (None, None, None, None, None) + tuple(gx) + tuple(gw)


class SSDLoss(nn.Module):
    """losses.py:159-315.  ctor `(iou_boxes, cls_criterion, coord_criterion, num_classes, img_size)`;
    forward((predicted_offsets [B,P,4], predicted_classes [B,P,C]), targets) -> {"loss","Localization","Classification"}.
    The reference's quirk is kept: image j is matched against the first lens[j] rows of `targets`."""

    def __init__(self, iou_boxes, cls_criterion, coord_criterion, num_classes, img_size):
        super().__init__()
        self.num_classes = num_classes
        self.img_size = img_size
        self.default_boxes = iou_boxes
        name = getattr(cls_criterion, "__name__", cls_criterion)
        if cls_criterion is focal_loss or name == "focal_loss":
            self._cls_mode = _lib.CLS_SOFTMAX_FOCAL
        elif cls_criterion is nn.CrossEntropyLoss or name in ("ce_loss", "CrossEntropyLoss"):
            self._cls_mode = _lib.CLS_MODES["ce_loss"]
        else:
            self._cls_mode = None        # nn.BCELoss(reduction='none') fails at the first forward in the reference
        self._coord_name = _coord_name(coord_criterion)
        self._ws = None
        self._priors_dev = None
        self.last_counts = None

    def _launch(self, off, cls, g_off, g_cls, target, rows_only=None):
        if self._cls_mode is None:
            raise ValueError("Using a target size (torch.Size([P])) that is different to the input size "
                             "(torch.Size([P, C])) is deprecated. Please ensure they have the same size.")
        _lib.require_cuda(off, cls, target)
        L = _lib.lib()
        dev = off.device
        if self._priors_dev is None or self._priors_dev.device != dev:
            self._priors_dev = torch.as_tensor(self.default_boxes, dtype=torch.float32).to(dev).contiguous()
        pri = self._priors_dev
        B, Pn, C = cls.shape
        if off.shape != (B, Pn, 4) or pri.shape != (Pn, 4):
            raise ValueError(f"shape mismatch: offsets {tuple(off.shape)}, classes {tuple(cls.shape)}, priors {tuple(pri.shape)}")
        tgt = target.float().contiguous()
        d = _lib.SsdLossDesc()
        d.off, d.cls = off.data_ptr(), cls.data_ptr()
        d.grad_off = g_off.data_ptr() if g_off is not None else None
        d.grad_cls = g_cls.data_ptr() if g_cls is not None else None
        d.priors, d.target = pri.data_ptr(), tgt.data_ptr()
        d.B, d.P, d.C, d.nT = B, Pn, C, tgt.shape[0]
        d.cls_mode, d.coord_mode = self._cls_mode, _lib.COORD_MODES[self._coord_name]
        if rows_only is not None:
            d.flags, d.active, d.active_count = _lib.SSD_ROWS_ONLY, rows_only[0].data_ptr(), rows_only[1].data_ptr()
        need = L.odpl_ssd_loss_workspace_bytes(ctypes.byref(d))
        if need == 0:
            raise OdplError("odpl_ssd_loss_workspace_bytes rejected the descriptor (empty targets raise in the reference too)")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
        out3 = torch.empty(3, dtype=torch.float32, device=dev)
        counts = torch.empty(2, dtype=torch.int32, device=dev)
        _lib.check(L.odpl_ssd_loss(ctypes.byref(d), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(out3), _lib.ptr(counts),
                                   _lib.current_stream(dev)), "odpl_ssd_loss")
        self.last_counts = counts
        return out3, counts

    fuse_head_backward = True      # class switch: False = always hand autograd the dense d loss/d preds
    # the structured route pays only while the gradient is sparse (utils/ssdhead.py: 2 % slower at 2 boxes per image, 2 % faster at 1, 11 % slower
    # at 8): taken when the batch holds at most this many target rows per image (a host-side number: no sync); None = whenever possible
    structured_max_boxes_per_image = 1.5

    def forward(self, outputs, targets):
        from .utils import ssdhead
        sparse = (self.structured_max_boxes_per_image is None
                  or targets.shape[0] <= self.structured_max_boxes_per_image * outputs[0].shape[0])
        src = ssdhead.fusable_source(outputs[0], outputs[1]) if self.fuse_head_backward and sparse and self._cls_mode is not None else None
        if src is not None:
            loss, m, _ = _SSDLossHeadFn.apply(self, targets, outputs[0].detach(), outputs[1].detach(), src, *src["xs"], *src["w"])
            return {"loss": loss, "Localization": m[0], "Classification": m[1]}
        predicted_offsets, predicted_classes = (_widen(t) for t in outputs)
        assert predicted_classes.size(0) == predicted_offsets.size(0)
        loss, m, _ = _SSDLossFn.apply(self, targets, predicted_offsets, predicted_classes)
        return {"loss": loss, "Localization": m[0], "Classification": m[1]}


def _v5_layout(h):
    """-> (tensor handed to the library, ODPL layout) for a [B, na, G, G, no] head.  The contiguous tensor (the reference's
    permute().contiguous()) and the permuted VIEW of the Detect convolution's NCHW output are both consumed in place."""
    if h.is_contiguous():
        return h, _lib.LAYOUT_INTERLEAVED
    B, na, G, G2_, no = h.shape
    if h.dim() == 5 and h.stride() == (na * no * G * G2_, no * G * G2_, G2_, 1, G * G2_):
        return h, _lib.LAYOUT_PLANAR
    return h.contiguous(), _lib.LAYOUT_INTERLEAVED


class _V5LossFn(torch.autograd.Function):
    """forward: loss + the compact gradient; backward: one expand launch (+ the matched cells) into dense tensors with the
    heads' own strides, scaled by the upstream gradient on the device."""

    @staticmethod
    def forward(ctx, owner, target, *heads):
        need = any(ctx.needs_input_grad[2:])
        pairs = [_v5_layout(h) for h in heads]
        d, keep = owner._describe([p[0] for p in pairs], [p[1] for p in pairs], target)
        out4, counts, compact = owner._forward(d, pairs[0][0].device, need)
        ctx.desc, ctx.keep, ctx.compact = d, keep, compact
        ctx.like = [p[0] for p in pairs] if need else None
        ctx.used = False
        metrics = out4[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out4[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        if ctx.used:
            raise RuntimeError("the fused YOLOv5 loss frees its gradient buffers after the first backward")
        ctx.used = True
        d, compact, like = ctx.desc, ctx.compact, ctx.like
        ctx.desc = ctx.compact = ctx.like = ctx.keep = None
        if not all(ctx.needs_input_grad[2:]):
            raise RuntimeError("the YOLOv5 criterion differentiates all of its heads or none")
        grads = [torch.empty_like(h) for h in like]                           # preserves the (possibly permuted-view) strides
        for i, g in enumerate(grads):
            d.layer[i].grad = g.data_ptr()
        up = g_loss.contiguous().float()
        _lib.check(_lib.lib().odpl_yolov5_loss_backward(ctypes.byref(d), _lib.ptr(compact), compact.numel(), _lib.ptr(up),
                                                        _lib.current_stream(like[0].device)), "odpl_yolov5_loss_backward")
        return (None, None) + tuple(grads)


class _V5LossHeadFn(torch.autograd.Function):
    """The same loss when every head is a view of a `utils.headconv.head_conv` output: the differentiable inputs are the Detect
    convolutions' (x, weight, bias); their backward runs from the compact gradient (`odpl_head_conv_backward_multi`)."""

    @staticmethod
    def forward(ctx, owner, target, n, *args):
        heads, srcs = args[:n], args[n:]
        d, keep = owner._describe(list(heads), [_lib.LAYOUT_PLANAR] * n, target)
        out4, counts, compact = owner._forward(d, heads[0].device, True)
        ctx.desc, ctx.keep, ctx.compact, ctx.n = d, keep, compact, n
        ctx.has = [t is not None for t in srcs]        # see losses._RegionLossHeadFn: version-checked by autograd
        ctx.save_for_backward(*[t for t in srcs if t is not None])
        ctx.used = False
        metrics = out4[1:]
        ctx.mark_non_differentiable(metrics, counts)
        return out4[0], metrics, counts

    @staticmethod
    def backward(ctx, g_loss, _gm, _gc):
        from .utils.headconv import head_conv_backward
        if ctx.used:
            raise RuntimeError("the fused YOLOv5 loss frees its gradient buffers after the first backward")
        ctx.used = True
        d, compact, n = ctx.desc, ctx.compact, ctx.n
        saved = iter(ctx.saved_tensors)
        srcs = [next(saved) if h else None for h in ctx.has]
        xs, ws, bs = srcs[0::3], srcs[1::3], srcs[2::3]
        ctx.desc = ctx.compact = ctx.keep = None
        up = g_loss.contiguous().float()
        L = _lib.lib()
        sparses = []
        for i in range(n):
            sg = _lib.SparseGrad()
            _lib.check(L.odpl_yolov5_compact_view(ctypes.byref(d), _lib.ptr(compact), i, ctypes.byref(sg), None), "odpl_yolov5_compact_view")
            sparses.append(sg)
        need_all = ctx.needs_input_grad[3 + n:]
        out = head_conv_backward(sparses, xs, ws, bs, up, [need_all[3 * i:3 * i + 3] for i in range(n)])
        return (None, None, None) + (None,) * n + tuple(out)


class _PosWeight(nn.Module):
    def __init__(self):
        super().__init__()
        self.register_buffer("pos_weight", torch.tensor([1.0]))


class _FocalShell(nn.Module):
    """Parameter-free stand-in for FocalLoss(nn.BCEWithLogitsLoss(pos_weight=[1.0])): keeps the buffer keys only;
    gamma 1.5 / alpha 0.25 are evaluated inside the HIP kernels."""

    def __init__(self, gamma=1.5, alpha=0.25):
        super().__init__()
        self.loss_fcn = _PosWeight()
        self.gamma, self.alpha, self.reduction = gamma, alpha, "mean"


class MultiScaleRegionLoss_v5(nn.Module):
    """losses.py:70-156.  ctor keeps the reference's signature (the three criterion arguments are unused there
    as well: BCE-with-logits + focal gamma 1.5 are hard-wired, losses.py:78-89).  `iou_mode` selects the
    bbox_iou_v5 branch; the reference always calls GIoU (losses.py:118), CIoU/DIoU exist in accuracy.py:101-112.
    forward(output, target) with output = list of [B, na, G, G, 5+nc] tensors -> dict(loss, Localization,
    Classification, Conf_obj)."""

    def __init__(self, anchors, anch_masks, cls_criterion, coord_criterion, conf_criterion, num_classes, img_dim=416,
                 iou_mode="giou", **kwargs):
        super().__init__()
        self.nc = num_classes
        self.nl = len(anchors)
        self.na = len(anchors[0]) // 2
        self.stride = torch.tensor([8., 16., 32.])                                  # losses.py:73
        self.anchor_grid = torch.tensor(anchors).float().view(self.nl, 1, -1, 1, 1, 2)
        self.anchors = self.anchor_grid.view(self.nl, -1, 2) / self.stride[:self.nl].view(-1, 1, 1)   # :95-96 (f32)
        self.red = "mean"
        # state_dict parity: the reference's criterion holds two FocalLoss(BCEWithLogitsLoss(pos_weight=[1.0])) modules
        # (losses.py:79-89) whose `pos_weight` buffers appear in the model's state_dict
        self.BCEcls, self.BCEobj = _FocalShell(), _FocalShell()
        self.iou_mode = iou_mode
        self._ws = None
        self._clean_key = None
        self.last_counts = None
        self.last_compact = (None, None)

    fuse_head_backward = True      # class switch: False = always hand autograd the dense d loss/d p

    def _describe(self, heads, layouts, target):
        _lib.require_cuda(target, *heads)
        d = _lib.YoloV5LossDesc()
        tgt = target.float().contiguous()
        d.nl, d.B, d.na, d.nc, d.nT = len(heads), heads[0].shape[0], self.na, self.nc, tgt.shape[0]
        d.iou_mode = _lib.IOU_MODES[self.iou_mode]
        d.bf16 = 1 if heads[0].dtype == torch.bfloat16 else 0
        d.target = tgt.data_ptr() if d.nT else None
        for i, (h, lay) in enumerate(zip(heads, layouts)):
            if h.dtype not in (torch.float32, torch.bfloat16) or h.dtype != heads[0].dtype:
                raise TypeError("YOLOv5 heads must all be fp32 or all be bf16 (arithmetic and sums are fp32 either way)")
            if h.dim() != 5 or h.shape[1] != self.na or h.shape[4] != 5 + self.nc or h.shape[2] != h.shape[3]:
                raise ValueError(f"layer {i}: expected [B,{self.na},G,G,{5 + self.nc}], got {tuple(h.shape)}")
            l = d.layer[i]
            l.p, l.grad, l.G, l.layout = h.data_ptr(), None, h.shape[2], lay
            for a in range(self.na):
                l.anchor_w[a], l.anchor_h[a] = float(self.anchors[i, a, 0]), float(self.anchors[i, a, 1])
        return d, (tgt, heads)

    def _forward(self, d, dev, need_grad):
        L = _lib.lib()
        need = L.odpl_yolov5_loss_workspace_bytes(ctypes.byref(d))
        if need == 0:
            raise OdplError("odpl_yolov5_loss_workspace_bytes rejected the descriptor")
        if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
            self._ws = torch.empty(int(need * 1.25) + 4096, dtype=torch.uint8, device=dev)
            self._clean_key = None
        key = (dev, torch.cuda.current_stream(dev).cuda_stream, d.B, tuple(d.layer[i].G for i in range(d.nl)))
        capturing = torch.cuda.is_current_stream_capturing()           # see losses.py: never claim STATE_CLEAN while recording
        d.flags = _lib.YOLOV5_STATE_CLEAN if (self._clean_key == key and not capturing) else 0
        if not capturing:
            self._clean_key = None
        compact = torch.empty(L.odpl_yolov5_loss_compact_bytes(ctypes.byref(d)), dtype=torch.uint8, device=dev) if need_grad else None
        out4 = torch.empty(4, dtype=torch.float32, device=dev)
        counts = torch.empty(_lib.MAX_SCALES + 1, dtype=torch.int32, device=dev)
        _lib.check(L.odpl_yolov5_loss_forward(ctypes.byref(d), _lib.ptr(self._ws), self._ws.numel(), _lib.ptr(compact),
                                              0 if compact is None else compact.numel(), _lib.ptr(out4), _lib.ptr(counts),
                                              _lib.current_stream(dev)), "odpl_yolov5_loss_forward")
        if not capturing:
            self._clean_key = key
        self.last_counts = counts
        self.last_compact = (d, compact)
        return out4, counts, compact

    def assignment_rows(self, layer):
        """build_targets_v5's output for `layer` as the kernels produced it in the last differentiable forward:
        (b, a, gj, gi) int64 [n], tbox f32 [n,4], anchor index int64 [n], target row int64 [n] (accuracy.py:472-521)."""
        d, compact = self.last_compact
        if compact is None:
            raise OdplError("assignment_rows needs a forward that kept its compact gradient (requires_grad heads)")
        rows = _lib.YoloV5Rows()
        _lib.check(_lib.lib().odpl_yolov5_compact_view(ctypes.byref(d), _lib.ptr(compact), layer, None, ctypes.byref(rows)), "odpl_yolov5_compact_view")
        base = compact.data_ptr()

        def view(ptr, n, dtype):
            off = ptr - base
            return compact[off:off + n * 4].view(dtype)
        n = int(view(rows.n_rows, 1, torch.int32)[0])
        cell = view(rows.row_cell, rows.cap, torch.int32)[:n].long()
        ag = view(rows.row_anchor_group, rows.cap, torch.int32)[:n].long()
        t = view(rows.row_target, rows.cap, torch.int32)[:n].long()
        tbox = view(rows.row_tbox, rows.cap * 4, torch.float32).view(-1, 4)[:n]
        G = d.layer[layer].G
        gi, gj = cell % G, (cell // G) % G
        plane = cell // (G * G)
        return (plane // self.na, plane % self.na, gj, gi), tbox, ag >> 4, t, cell

    def forward(self, output, target, seen=None):
        from .utils.headconv import fusable_source, structured_backward_supported
        output = list(output)
        srcs = []
        if self.fuse_head_backward:
            for h in output:                                              # a head is a permuted view of a head_conv output
                base = getattr(h, "_odpl_head_base", None)
                src = fusable_source(base) if base is not None else None
                srcs.append(src if (src is not None and _v5_layout(h)[1] == _lib.LAYOUT_PLANAR and h.data_ptr() == base.data_ptr()) else None)
        fuse = bool(srcs) and all(s is not None for s in srcs)
        if fuse:        # the structured head backward's own limits (odpl_yolov5_compact_view reports 5 * na * nT row slots per layer)
            nT = target.shape[0] if target.dim() == 2 else -1
            geoms = [(h.shape[0], h.shape[2] * h.shape[3], self.na, 5 + self.nc, 4, 5 * self.na * nT, _lib.LAYOUT_PLANAR) for h in output]
            fuse = nT >= 0 and structured_backward_supported(srcs, geoms)
        if fuse:
            flat = [t for s in srcs for t in s]
            loss, m, _ = _V5LossHeadFn.apply(self, target, len(output), *[h.detach() for h in output], *flat)
        else:
            loss, m, _ = _V5LossFn.apply(self, target, *output)
        return {"loss": loss, "Localization": m[0], "Classification": m[1], "Conf_obj": m[2]}


def configure(args, iou_boxes, anchors, anch_masks, num_classes, img_size):
    """The RetinaNet / SSD / YOLOv5 arms of configure_loss (losses.py:23-28)."""
    cls = {"focal_loss": focal_loss, "ce_loss": nn.CrossEntropyLoss, "bce_loss": nn.BCELoss}[args.cls_criterion]
    coord = {"mse_loss": nn.MSELoss, "smooth_l1_loss": nn.SmoothL1Loss}[args.coord_criterion]
    if args.model_name == "RetinaNet":
        return RetinaNetLoss(iou_boxes, cls, coord, num_classes, img_size)
    if args.model_name == "SSD":
        return SSDLoss(iou_boxes, cls, coord, num_classes, img_size)
    if args.model_name == "YOLOv5":
        return MultiScaleRegionLoss_v5(anchors, anch_masks, cls, coord, nn.BCELoss, num_classes, img_size)
    raise NotImplementedError(f"{args.model_name}: criterion not built yet (DESIGN.md section 7)")
