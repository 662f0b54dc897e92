"""SSD's six class / box head pairs (`model/SSD.py:249-310`: `cl{i}` / `bbx{i}` = Conv2d(Cin_i, n_i * C | n_i * 4, 3, padding=1) on the six
source maps, each followed by permute(0, 2, 3, 1) + view(B, -1, no) and a cat over the sources), hosted so that SSDLoss can reach THROUGH
them -- what `utils/retinahead.py` does for RetinaNet's shared heads, with per-source weights, channel counts and prior counts.

forward   the MIOpen convolutions per source and the reference's permute / view / cat: `out_bbx [B, 8732, 4]`, `out_cl [B, 8732, C]`.
backward  (a) from dense d loss/d preds (any consumer): aten::convolution_backward per source, as autograd would;
          (b) from the criterion: the outputs remember their producer (`_odpl_ssd_src`); SSDLoss writes its gradient on the positives and
          the mined hard negatives only (`ODPL_SSD_ROWS_ONLY`: d loss/d preds is zero on every other prior, losses.py:241-268),
          `odpl_ssd_level_lists` cuts that list by source and `odpl_retina_heads_backward` -- one call per source, both heads on the one
          shared input -- computes dX, dW, db from those rows: the dense [B, n_i * C, H, W] gradients are never multiplied by.
"""
import ctypes

import torch
import torch.nn.functional as F

from ... import _lib

# Switch for model/SSD.py: True = host the head pairs (this module), so that SSDLoss MAY take the structured route; the dense route through
# the hosted node is the same aten::convolution_backward calls autograd makes for the plain modules.  Which route a step takes is
# SSDLoss's choice (`SSDLoss.structured_max_boxes_per_image`), because the structured one only pays on sparse gradients: measured on SSD300
# (B = 32, fp32, tools/bench_ssd_step.py, profiles/r04_ssd_step_ab.jsonl) the loss's gradient lives on positives + 3:1 mined negatives --
# ~600 priors per image with 8 boxes (7 % of 8 732; ~19 000 rows per step), and from that many rows the gathered VALU products
# (rh_dw / rh_dx / rh_pair: 8.6 ms) cost more than MIOpen's dense MFMA backward of the twelve convolutions: 61.6 vs 55.4 ms per step at
# 8 boxes per image, 56.2 vs 55.3 at 2, 54.4 vs 55.3 ms at 1 (with SSD's convolution problems in the committed MIOpen db; before they
# were: 73.1 vs 67.4, a tie at 2, 65.7 vs 67.2).  RetinaNet's heads (863 rows of 802 800, config 4) are where the same kernels win outright.
enabled = True


class _HeadPairsFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, n, C, *t):
        xs, ws = t[:n], t[n:]                                  # ws: (w_cl, b_cl, w_bbx, b_bbx) per source
        B = xs[0].shape[0]
        cl, bbx = [], []
        for i, x in enumerate(xs):
            wc, bc, wb, bb = ws[4 * i:4 * i + 4]
            cl.append(F.conv2d(x, wc, bc, padding=1).permute(0, 2, 3, 1).reshape(B, -1, C))
            bbx.append(F.conv2d(x, wb, bb, padding=1).permute(0, 2, 3, 1).reshape(B, -1, 4))
        ctx.save_for_backward(*t)
        ctx.n, ctx.C = n, C
        return torch.cat(bbx, 1), torch.cat(cl, 1)

    @staticmethod
    def backward(ctx, g_bbx, g_cl):
        t = ctx.saved_tensors
        n, C = ctx.n, ctx.C
        xs, ws = t[:n], t[n:]
        B = xs[0].shape[0]
        gxs, gws = [], []
        off = 0
        for i, x in enumerate(xs):
            wc, bc, wb, bb = ws[4 * i:4 * i + 4]
            H, W = x.shape[2], x.shape[3]
            A = wb.shape[0] // 4
            rows = H * W * A
            gx = None
            for w, g, no in ((wc, g_cl, C), (wb, g_bbx, 4)):
                if g is None:
                    gws += [None, None]
                    continue
                gl = g[:, off:off + rows, :].reshape(B, H, W, A * no).permute(0, 3, 1, 2)            # channels-last [B, A * no, H, W]
                gxi, gw, gb = torch.ops.aten.convolution_backward(gl, x, w, [A * no], [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, True, True])
                gx = gxi if gx is None else gx + gxi
                gws += [gw, gb]
            gxs.append(gx)
            off += rows
        return (None, None) + tuple(gxs) + tuple(gws)


def is_hostable(conv):
    return (isinstance(conv, torch.nn.Conv2d) and conv.kernel_size == (3, 3) and conv.stride == (1, 1) and conv.padding == (1, 1)
            and conv.dilation == (1, 1) and conv.groups == 1 and conv.padding_mode == "zeros" and conv.bias is not None)


def head_pairs(cl_convs, bbx_convs, sources, num_classes):
    """-> (out_bbx, out_cl) as SSD.forward returns them; the outputs carry what SSDLoss needs to run the twelve convolutions' backward."""
    n = len(sources)
    flat = []
    for c, b in zip(cl_convs, bbx_convs):
        flat += [c.weight, c.bias, b.weight, b.bias]
    out_bbx, out_cl = _HeadPairsFn.apply(n, num_classes, *sources, *flat)
    src = {"xs": tuple(sources), "w": tuple(flat), "n": n, "C": num_classes, "versions": (out_bbx._version, out_cl._version)}
    out_bbx._odpl_ssd_src = src
    out_cl._odpl_ssd_src = src
    return out_bbx, out_cl


def _query(B, A, cin, H, W, C, dev_ptr=1):
    """workspace bytes of one source's call (0 = outside the library's limits); pointers only need to be non-NULL"""
    L = _lib.lib()
    d = _lib.RetinaHeadsDesc()
    d.B, d.A, d.Cin, d.nlevels = B, A, cin, 1
    d.H[0], d.W[0] = H, W
    d.pos_anchor = d.num_pos = dev_ptr
    d.rows_per_image, d.flags = 0, _lib.HEADS_SHARED_INPUT
    Hd = (_lib.RetinaHead * 2)()
    for h, no in zip(Hd, (4, C)):
        h.x[0] = h.grad_x[0] = dev_ptr
        h.w = h.grad_rows = dev_ptr
        h.w_stride[0], h.w_stride[1], h.w_stride[2], h.w_stride[3] = cin * 9, 9, 3, 1
        h.no = no
    return L.odpl_retina_heads_backward_workspace_bytes(ctypes.byref(d), 2, Hd)


def fusable_source(out_bbx, out_cl):
    """The producer record when SSDLoss may run the head convolutions' backward itself, else None: both tensors come from ONE `head_pairs`
    call, untouched since, nothing hangs on their autograd nodes, fp32 channels-last sources on the device within the library's limits."""
    src = getattr(out_bbx, "_odpl_ssd_src", None)
    if src is None or getattr(out_cl, "_odpl_ssd_src", None) is not src or not torch.is_grad_enabled():
        return None
    if (out_bbx._version, out_cl._version) != src["versions"]:
        return None
    for t in (out_bbx, out_cl):
        if t.retains_grad or getattr(t, "_backward_hooks", None) or t.dtype != torch.float32 or not t.is_contiguous():
            return None
    B = src["xs"][0].shape[0]
    for i, x in enumerate(src["xs"]):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == B and x.shape[1] % 4 == 0
                and x.is_contiguous(memory_format=torch.channels_last) and x.data_ptr() % 16 == 0):
            return None
        wc, bc, wb, bb = src["w"][4 * i:4 * i + 4]
        for w in (wc, bc, wb, bb):
            if w is None or w.dtype != torch.float32 or not w.is_cuda:
                return None
        if _query(B, wb.shape[0] // 4, x.shape[1], x.shape[2], x.shape[3], src["C"]) == 0:
            return None
    if not any(t.requires_grad for t in src["xs"] + src["w"]):
        return None
    return src


def level_first(src):
    """first prior of every source inside the 8732-row tensors (+ the total), from the sources' shapes and the box heads' widths"""
    first = [0]
    for i, x in enumerate(src["xs"]):
        first.append(first[-1] + x.shape[2] * x.shape[3] * (src["w"][4 * i + 2].shape[0] // 4))
    return first


def heads_backward(src, g_bbx, g_cl, lists, counts, upstream, needs_x, needs_w):
    """`odpl_retina_heads_backward`, one call per source.  g_bbx / g_cl: the criterion's gradients in the dense [B, 8732, no] layout with only
    the active rows written; lists / counts: `odpl_ssd_level_lists`.  -> ([gx per source], [gw_cl, gb_cl, gw_bbx, gb_bbx per source])"""
    L = _lib.lib()
    xs, ws, n, C = src["xs"], src["w"], src["n"], src["C"]
    dev = xs[0].device
    B, P = g_bbx.shape[0], g_bbx.shape[1]
    first = level_first(src)
    st = _lib.current_stream(dev)
    out_x, out_w = [], []
    wsb = None
    for i, x in enumerate(xs):
        wc, bc, wb, bb = ws[4 * i:4 * i + 4]
        A = wb.shape[0] // 4
        d = _lib.RetinaHeadsDesc()
        d.B, d.A, d.Cin, d.nlevels = B, A, x.shape[1], 1
        d.H[0], d.W[0] = x.shape[2], x.shape[3]
        d.pos_anchor = lists.data_ptr() + 4 * B * first[i]
        d.num_pos = counts.data_ptr() + 4 * i
        d.upstream = upstream.data_ptr() if upstream is not None else None
        d.rows_per_image, d.flags = P, _lib.HEADS_SHARED_INPUT
        gx = torch.empty_like(x) if needs_x[i] else None                        # channels-last, like x
        Hd = (_lib.RetinaHead * 2)()
        grads = []
        for h, (w, g, no, nw) in zip(Hd, ((wb, g_bbx, 4, needs_w[4 * i + 2:4 * i + 4]), (wc, g_cl, C, needs_w[4 * i:4 * i + 2]))):
            h.x[0] = x.data_ptr()
            h.grad_x[0] = gx.data_ptr() if gx is not None else None
            gw = torch.empty_like(w) if nw[0] else None
            gb = torch.empty(w.shape[0], dtype=torch.float32, device=dev) if nw[1] else None
            h.w, h.grad_w, h.grad_b = w.data_ptr(), (gw.data_ptr() if gw is not None else None), (gb.data_ptr() if gb is not None else None)
            h.grad_rows = g.data_ptr() + 4 * first[i] * no                       # the source's first prior; rows_per_image strides the images
            for k in range(4):
                h.w_stride[k] = w.stride(k)
            h.no = no
            grads.append((gw, gb))
        nbytes = L.odpl_retina_heads_backward_workspace_bytes(ctypes.byref(d), 2, Hd)
        if nbytes == 0:
            raise _lib.OdplError("odpl_retina_heads_backward_workspace_bytes rejected an SSD source (fusable_source admitted it)")
        if wsb is None or wsb.numel() < nbytes:
            wsb = torch.empty(nbytes, dtype=torch.uint8, device=dev)             # one buffer, reused source after source (stream order)
        _lib.check(L.odpl_retina_heads_backward(ctypes.byref(d), 2, Hd, _lib.ptr(wsb), wsb.numel(), st), "odpl_retina_heads_backward")
        out_x.append(gx)
        out_w += [grads[1][0], grads[1][1], grads[0][0], grads[0][1]]             # (w_cl, b_cl, w_bbx, b_bbx)
    return out_x, out_w
