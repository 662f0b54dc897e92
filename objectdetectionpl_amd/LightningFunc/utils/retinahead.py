"""RetinaNet's two final head convolutions (`model/RetinaNet.py:93-113`: `loc_head[-1]`, `cls_head[-1]`, shared by the five FPN levels,
followed by permute(0, 2, 3, 1) + view(B, -1, no) + cat over the levels), hosted so that RetinaNetLoss can reach THROUGH them.

forward   the MIOpen convolutions per level, the reference's permute / view / cat: `loc_preds [B, An, 4]`, `cls_preds [B, An, C]`.
backward  (a) from dense d loss/d preds (any consumer): aten::convolution_backward per level, as autograd would;
          (b) from the criterion: the outputs remember their producer (`_odpl_retina_src`); RetinaNetLoss writes its gradient on the
          positive rows only and `odpl_retina_heads_backward` (csrc/odpl_retinagrad.hip) computes dX, dW, db of both convolutions on
          all levels from those ~0.1 % of the rows: the dense [B, 9 * C, H, W] gradients are never multiplied by.
"""
import ctypes

import torch
import torch.nn.functional as F

from ... import _lib

enabled = True          # A/B switch (tools/bench_retina_step.py --no-hosted-heads): False = the plain nn.Conv2d modules + cat


class PixelSupport:
    """Per level l and reach j: the pixels within Chebyshev distance j of a pixel that owns a positive anchor (`odpl_pixel_support`),
    as device lists.  `pair(l, j)` -> (list, count) for `sparseconv.forward_on_pixels`; `has(l)`: lists exist for that level."""

    def __init__(self, pos, npos, B, shapes, depth, mask, A=9):
        L = _lib.lib()
        dev = pos.device
        self.B, self.depth, self.mask, self.shapes = B, depth, mask, list(shapes)
        pix = [B * h * w for h, w in shapes]
        self.base = [depth * sum(pix[:i]) for i in range(len(pix))]
        self.pix = pix
        self.lists = torch.empty(depth * sum(pix), dtype=torch.int32, device=dev)
        self.counts = torch.empty(len(pix) * depth, dtype=torch.int32, device=dev)
        d = _lib.PixelSupport()
        d.pos_anchor, d.num_pos = pos.data_ptr(), npos.data_ptr()
        d.B, d.An, d.A, d.n_levels = B, A * sum(h * w for h, w in shapes), A, len(pix)
        for i, (h, w) in enumerate(shapes):
            d.H[i], d.W[i] = h, w
        d.depth, d.level_mask = depth, mask
        d.lists, d.counts = self.lists.data_ptr(), self.counts.data_ptr()
        n = L.odpl_pixel_support_workspace_bytes(ctypes.byref(d))
        if n == 0:
            raise _lib.OdplError("odpl_pixel_support_workspace_bytes rejected the descriptor")
        ws = torch.empty(n, dtype=torch.uint8, device=dev)
        _lib.check(L.odpl_pixel_support(ctypes.byref(d), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)), "odpl_pixel_support")

    def has(self, l):
        return bool((self.mask >> l) & 1)

    def pair(self, l, j):
        o = self.base[l] + j * self.pix[l]
        return self.lists[o:o + self.pix[l]], self.counts[l * self.depth + j:l * self.depth + j + 1]


class _FinalConvsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, wl, bl, wc, bc, n, support, *xs):
        B = xs[0].shape[0]
        outs = []
        for w, b, group in ((wl, bl, xs[:n]), (wc, bc, xs[n:])):
            no = w.shape[0] // 9
            if support is None:
                outs.append(torch.cat([F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(B, -1, no) for x in group], 1))
                continue
            # loss-driven step: on the levels with pixel lists only the pixels that own a positive anchor are evaluated (the rows the
            # criterion reads); the other rows of those levels stay zero
            from .sparseconv import forward_on_pixels
            An = 9 * sum(x.shape[2] * x.shape[3] for x in group)
            out = torch.zeros((B, An, no), dtype=torch.float32, device=xs[0].device)
            first = 0
            for l, x in enumerate(group):
                rows = 9 * x.shape[2] * x.shape[3]
                if support.has(l):
                    forward_on_pixels(x, w, b, 0, support.pair(l, 0), out=out[:, first:first + rows], out_strides=(An * no, 9 * no))
                else:
                    out[:, first:first + rows] = F.conv2d(x, w, b, padding=1).permute(0, 2, 3, 1).reshape(B, rows, no)
                first += rows
            outs.append(out)
        ctx.save_for_backward(wl, bl, wc, bc, *xs)
        ctx.n = n
        return outs[0], outs[1]

    @staticmethod
    def backward(ctx, g_loc, g_cls):
        wl, bl, wc, bc, *xs = ctx.saved_tensors
        n = ctx.n
        B = xs[0].shape[0]
        grads = {}
        gxs = []
        for name, w, b, group, g in (("l", wl, bl, xs[:n], g_loc), ("c", wc, bc, xs[n:], g_cls)):
            co = w.shape[0]
            gw = gb = None
            off = 0
            for x in group:
                H, W = x.shape[2], x.shape[3]
                rows = H * W * 9
                if g is None:
                    gxs.append(None)
                    off += rows
                    continue
                gl = g[:, off:off + rows, :].reshape(B, H, W, co).permute(0, 3, 1, 2)        # channels-last [B, co, H, W]
                off += rows
                gx, gwl, gbl = torch.ops.aten.convolution_backward(gl, x, w, [co], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
                gxs.append(gx)
                gw = gwl if gw is None else gw + gwl
                gb = gbl if gb is None else gb + gbl
            grads[name] = (gw, gb)
        return (grads["l"][0], grads["l"][1], grads["c"][0], grads["c"][1], None, None) + tuple(gxs)


def final_convs(loc_conv, cls_conv, xl, xc, support=None, assignment=None):
    """-> (loc_preds, cls_preds); the outputs carry what the criterion needs to run both convolutions' backward itself.
    support (PixelSupport) + assignment (RetinaNetLoss.assign's token): the loss-driven step -- predictions exist on the positive
    pixels of the listed levels only, and ONLY the criterion that made the assignment may consume them."""
    n = len(xl)
    loc, cls = _FinalConvsFn.apply(loc_conv.weight, loc_conv.bias, cls_conv.weight, cls_conv.bias, n, support, *xl, *xc)
    src = {"w": (loc_conv.weight, loc_conv.bias, cls_conv.weight, cls_conv.bias), "xs": tuple(xl) + tuple(xc), "n": n,
           "versions": (loc._version, cls._version), "assignment": assignment}
    loc._odpl_retina_src = src
    cls._odpl_retina_src = src
    return loc, cls


def is_hostable(conv):
    return (conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1) and conv.dilation == (1, 1) and conv.groups == 1
            and conv.padding_mode == "zeros" and conv.bias is not None and conv.out_channels % 9 == 0)


def fusable_source(loc, cls):
    """The shared producer record when RetinaNetLoss may run the final convolutions' backward itself, else None: both tensors come
    from ONE `final_convs` call, untouched since (version counters), nothing hangs on their autograd nodes (hooks, retain_grad), fp32
    dense channels-last inputs on the device, gradients being recorded."""
    src = getattr(loc, "_odpl_retina_src", None)
    if src is None or getattr(cls, "_odpl_retina_src", None) is not src or not torch.is_grad_enabled():
        return None
    if (loc._version, cls._version) != src["versions"]:
        return None
    for t in (loc, cls):
        if t.retains_grad or getattr(t, "_backward_hooks", None) or t.dtype != torch.float32 or not t.is_contiguous():
            return None
    xs = src["xs"]
    cin = xs[0].shape[1]
    for x in xs:
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == cin and cin % 4 == 0 and cin <= 448
                and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0):
            return None
    for w in src["w"]:
        if w is None or w.dtype != torch.float32 or not w.is_cuda:
            return None
    if not any(t.requires_grad for t in xs + src["w"]):
        return None
    return src


def heads_backward(src, g_loc, g_cls, pos_anchor, num_pos, upstream, needs_w, needs_x):
    """`odpl_retina_heads_backward` for both heads.  g_loc / g_cls: the criterion's gradients in the dense [B, An, no] layout with
    only the rows of pos_anchor written.  needs_w: 4 booleans (wl, bl, wc, bc); needs_x: one per level input.
    -> ([gwl, gbl, gwc, gbc], [gx...])"""
    L = _lib.lib()
    wl, bl, wc, bc = src["w"]
    xs, n = src["xs"], src["n"]
    dev = xs[0].device
    B, cin = xs[0].shape[0], xs[0].shape[1]
    d = _lib.RetinaHeadsDesc()
    d.B, d.A, d.Cin, d.nlevels = B, 9, cin, n
    for i in range(n):
        d.H[i], d.W[i] = xs[i].shape[2], xs[i].shape[3]
    d.pos_anchor, d.num_pos = pos_anchor.data_ptr(), num_pos.data_ptr()
    d.upstream = upstream.data_ptr() if upstream is not None else None
    H = (_lib.RetinaHead * 2)()
    out_w, out_x = [], []
    for hi, (w, b, group, g, nw, nx) in enumerate(((wl, bl, xs[:n], g_loc, needs_w[0:2], needs_x[:n]), (wc, bc, xs[n:], g_cls, needs_w[2:4], needs_x[n:]))):
        h = H[hi]
        want_x = any(nx)
        gxs = [torch.empty_like(x) if want_x else None for x in group]           # channels-last, like x
        for i, (x, gx) in enumerate(zip(group, gxs)):
            h.x[i] = x.data_ptr()
            h.grad_x[i] = gx.data_ptr() if gx is not None else None
        gw = torch.empty_like(w) if nw[0] else None                               # the parameter's own strides
        gb = torch.empty(w.shape[0], dtype=torch.float32, device=dev) if nw[1] else None
        h.w, h.grad_w, h.grad_b = w.data_ptr(), (gw.data_ptr() if gw is not None else None), (gb.data_ptr() if gb is not None else None)
        h.grad_rows = g.data_ptr()
        for k in range(4):
            h.w_stride[k] = w.stride(k)
        h.no = w.shape[0] // 9
        out_w += [gw, gb]
        out_x += [gx if need else None for gx, need in zip(gxs, nx)]
    nbytes = L.odpl_retina_heads_backward_workspace_bytes(ctypes.byref(d), 2, H)
    if nbytes == 0:
        raise _lib.OdplError("odpl_retina_heads_backward_workspace_bytes rejected the descriptors")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    _lib.check(L.odpl_retina_heads_backward(ctypes.byref(d), 2, H, _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)), "odpl_retina_heads_backward")
    return out_w, out_x
