"""The linear 1x1 convolution that produces a detection head (`model/YOLOV4.py:89-127` conv2 / conv10 / conv18; the v2 / v3
heads; `model/YOLOV5.py:93-97` Detect.m), hosted so that the loss can reach THROUGH it.

forward   out[b] = W[Cout, Cin] @ X_b^T[Cin, G*G] + bias: one library batched GEMM on the channels-last activations that writes
          the head NCHW-planar -- the format every detector kernel wants (confidence rows contiguous), at the price of
          MIOpen's own 1x1 forward (tools/probe_headconv.py: 197 vs 189 us for the three YOLOv4 heads) and without the two
          layout copies a channels-last head costs (2 x 99 us).
backward  (a) from a dense d loss/d head (any consumer): three plain GEMMs, 307 us against MIOpen's 886 us;
          (b) from the criterion: the head tensor remembers its producer (`out._odpl_head_src`), the YOLO losses hand their
          STRUCTURED gradient (one value per cell + 4+C values per matched cell) straight to `odpl_head_conv_backward`
          (csrc/odpl_headgrad.hip) and the dense 57.9 MB tensor is never written, read or multiplied by.
"""
import ctypes

import torch
import torch.nn as nn

from ... import _lib


def _rows_view(x):
    """[B, Cin, H, W] -> ([B, H*W, Cin] view or None, True when x is dense channels-last)."""
    B, C, H, W = x.shape
    if x.is_contiguous(memory_format=torch.channels_last) and (C > 1 or not x.is_contiguous()):
        return x.permute(0, 2, 3, 1).reshape(B, H * W, C), True
    return None, False


class _HeadConv1x1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b):
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        wm = w.reshape(Cout, Cin)
        if wm.dtype != x.dtype:
            wm = wm.to(x.dtype)
        bias = b.to(x.dtype).view(1, Cout, 1) if b is not None else None
        rows, cl = _rows_view(x)
        xt = rows.transpose(1, 2) if cl else x.reshape(B, Cin, H * W)                 # [B, Cin, HW] (a view either way)
        if bias is not None:
            out = torch.baddbmm(bias, wm.unsqueeze(0).expand(B, Cout, Cin), xt)
        else:
            out = torch.matmul(wm, xt)
        ctx.save_for_backward(x, w)
        ctx.has_bias = b is not None
        return out.view(B, Cout, H, W)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        B, Cin, H, W = x.shape
        Cout = w.shape[0]
        gm = g.reshape(B, Cout, H * W)
        if gm.dtype != x.dtype:
            gm = gm.to(x.dtype)
        rows, cl = _rows_view(x)
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            wm = w.reshape(Cout, Cin).to(x.dtype)
            if cl:
                gx = torch.matmul(gm.transpose(1, 2), wm).view(B, H, W, Cin).permute(0, 3, 1, 2)      # channels-last, like x
            else:
                gx = torch.matmul(wm.t(), gm).view(B, Cin, H, W)
        if ctx.needs_input_grad[1]:
            xr = rows if cl else x.reshape(B, Cin, H * W).transpose(1, 2)
            gw = torch.bmm(gm, xr).sum(0).to(w.dtype).view_as(w)
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = gm.sum((0, 2), dtype=torch.float32)
        return gx, gw, gb


def head_conv(x, conv):
    """`conv(x)` for a 1x1, stride-1, ungrouped nn.Conv2d producing a detection head.  The result is NCHW-contiguous and
    remembers (x, weight, bias) so that a YOLO criterion can run the convolution's backward from its structured gradient."""
    out = _HeadConv1x1.apply(x, conv.weight, conv.bias)
    out._odpl_head_src = (x, conv.weight, conv.bias)
    out._odpl_head_version = out._version       # an in-place edit of the head between here and the criterion voids the shortcut
    return out


def is_head_conv(conv):
    return (conv.kernel_size == (1, 1) and conv.stride == (1, 1) and conv.padding == (0, 0) and conv.dilation == (1, 1)
            and conv.groups == 1 and conv.padding_mode == "zeros")


def fusable_source(h):
    """(x, w, b) when the criterion may run this head's convolution backward itself: fp32 (or bf16) dense channels-last
    input, fp32 weights, on the device, and gradients are being recorded.  The shortcut bypasses the head tensor's own autograd
    node, so it is refused when anything hangs on that node or the tensor no longer holds the convolution's output: a tensor
    hook, `retain_grad()`, or an in-place modification since `head_conv` (version counter)."""
    src = getattr(h, "_odpl_head_src", None)
    if src is None or not torch.is_grad_enabled():
        return None
    if h._version != getattr(h, "_odpl_head_version", -1) or h.retains_grad or getattr(h, "_backward_hooks", None):
        return None
    x, w, b = src
    if not (x.is_cuda and w.dtype == torch.float32 and x.dtype in (torch.float32, torch.bfloat16)):
        return None
    if _rows_view(x)[0] is None or x.shape[1] % (8 if x.dtype == torch.bfloat16 else 4):
        return None
    if not (x.requires_grad or w.requires_grad or (b is not None and b.requires_grad)):
        return None
    return src


def _call_groups(H, S, n):
    """the launches `head_conv_backward` makes for n heads: all of them in one multi call when anchors per head and dtype agree,
    one call per head otherwise -> list of (k, HeadConv array, SparseGrad array)"""
    same = all(S[i].A == S[0].A and H[i].bf16 == H[0].bf16 for i in range(n))
    if same:
        return [(n, H, S)]
    return [(1, (_lib.HeadConv * 1)(H[i]), (_lib.SparseGrad * 1)(S[i])) for i in range(n)]


def structured_backward_supported(srcs, geoms):
    """Asked in FORWARD, before the criterion commits to the structured route: does `odpl_head_conv_backward_multi` take these
    heads?  srcs: [(x, w, b)] per head; geoms: [(B, G2, A, no, dense_ch, n_rows, layout)] per head -- what the compact view will
    report.  The library's own validation answers (`*_workspace_bytes` == 0: more than 5 anchors on a scale, a geometry beyond its
    32-bit tile offsets, ...); the query reads no device memory, so the gradient pointers it does not have yet are stand-ins."""
    L = _lib.lib()
    n = len(srcs)
    if n < 1 or n > _lib.MAX_SCALES:
        return False
    H = (_lib.HeadConv * n)()
    S = (_lib.SparseGrad * n)()
    for i, ((x, w, _b), (B, G2, A, no, dc, n_rows, layout)) in enumerate(zip(srcs, geoms)):
        if x.data_ptr() % 16 or w.data_ptr() % 16:
            return False
        H[i].x, H[i].w, H[i].Cin, H[i].bf16 = x.data_ptr(), w.data_ptr(), x.shape[1], 1 if x.dtype == torch.bfloat16 else 0
        S[i].dense = S[i].rows = S[i].row_cell = x.data_ptr()          # non-NULL stand-ins (never dereferenced by the query)
        S[i].n_rows, S[i].B, S[i].G2, S[i].A, S[i].no, S[i].dense_ch, S[i].layout = n_rows, B, G2, A, no, dc, layout
    return all(L.odpl_head_conv_backward_multi_workspace_bytes(k, Hg, Sg) != 0 for k, Hg, Sg in _call_groups(H, S, n))


def head_conv_backward(sparses, xs, ws, bs, upstream, needs):
    """The backward of every head convolution of a model from the criterion's structured gradients: ONE
    `odpl_head_conv_backward_multi` call (two launches).  sparses: list of _lib.SparseGrad; needs[i] = (x, w, b) wanted.
    -> flat list [gx0, gw0, gb0, gx1, ...]."""
    L = _lib.lib()
    n = len(xs)
    dev = xs[0].device
    H = (_lib.HeadConv * n)()
    S = (_lib.SparseGrad * n)(*sparses)
    out = []
    for i, (x, w, b, need) in enumerate(zip(xs, ws, bs, needs)):
        gx = torch.empty_like(x) if need[0] else None                   # same (channels-last) strides as x
        gw = torch.empty_like(w) if need[1] else None                   # memory [Cout][Cin] in either memory format of a 1x1 weight
        gb = torch.empty(w.shape[0], dtype=torch.float32, device=dev) if (need[2] and b is not None) else None
        h = H[i]
        h.x, h.w = x.data_ptr(), w.data_ptr()
        h.grad_x = gx.data_ptr() if gx is not None else None
        h.grad_w = gw.data_ptr() if gw is not None else None
        h.grad_b = gb.data_ptr() if gb is not None else None
        h.upstream = upstream.data_ptr() if upstream is not None else None
        h.Cin = x.shape[1]
        h.bf16 = 1 if x.dtype == torch.bfloat16 else 0
        out += [gx, gw, gb]
    for k, Hg, Sg in _call_groups(H, S, n):
        nbytes = L.odpl_head_conv_backward_multi_workspace_bytes(k, Hg, Sg)
        if nbytes == 0:                 # cannot happen after structured_backward_supported() said yes in forward
            raise _lib.OdplError("odpl_head_conv_backward_multi_workspace_bytes rejected the descriptors")
        ws_buf = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        _lib.check(L.odpl_head_conv_backward_multi(k, Hg, Sg, _lib.ptr(ws_buf), ws_buf.numel(), _lib.current_stream(dev)),
                   "odpl_head_conv_backward_multi")
    for i, b in enumerate(bs):
        if out[3 * i + 2] is not None and b.dtype != torch.float32:
            out[3 * i + 2] = out[3 * i + 2].to(b.dtype)
    return out


class HeadConv2d(nn.Conv2d):
    """nn.Conv2d (same parameters, same state_dict keys) whose forward on the device goes through `head_conv`."""
    enabled = True          # class switch: False = the plain MIOpen convolution (heads then come out in the network's memory format)

    def forward(self, x):
        if HeadConv2d.enabled and x.is_cuda and x.dim() == 4 and is_head_conv(self):
            return head_conv(x, self)
        return super().forward(x)
