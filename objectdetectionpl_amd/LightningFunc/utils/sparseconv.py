"""`relu(conv3x3(x) + bias)` whose backward runs over the pixels that carry a gradient (RetinaNet's head towers, `model/RetinaNet.py:93-99`).

forward   the MIOpen convolution + the in-place bias + activation epilogue (`odpl_bias_act_forward`), as `bnact.conv_bias_act`.
backward  ONE autograd node for conv + bias + activation: `odpl_bias_act_backward_flags` turns grad_y into the convolution's output
          gradient, the bias gradient and a byte map of the pixels whose gradient row is non-zero; `odpl_sparse_conv3x3_backward`
          (csrc/odpl_sparseconv.hip) then computes dX and dW over those pixels only, as gathered GEMMs on the fp32 matrix cores.  Behind
          RetinaNetLoss the gradient spreads one pixel per layer from the positive anchors: a few per cent of the pixels of the two
          large FPN levels.  A dense gradient is still handled correctly (every pixel active), only slower than the library: the
          switch `enabled` and the pixel threshold `min_pixels` bound where it is used.
"""
import ctypes

import torch
import torch.nn.functional as F
from torch.utils.weak import WeakIdKeyDictionary

from ... import _lib
from .bnact import ACT, _ones_zeros, _workspace

enabled = True            # A/B switch (tools/bench_retina_step.py --dense-tower-backward)
min_pixels = 8192         # levels with fewer B*H*W pixels keep the library backward (they are 6 % of the towers' arithmetic at config 4)
max_density = 0.5         # above this share of pixels within reach of a gradient the library's dense backward is the faster one
reprobe_every = 200       # a layer sent to the dense route tries the sparse one again after this many calls
_ws2 = {}
_density = WeakIdKeyDictionary()    # weight -> {(H, W): {"d": last observed share, "pending": (pinned, event, pixels), "dense_calls": n}}


def _observe(key):
    """fold a finished asynchronous read-back of the pixel counts into the layer's record (never waits)"""
    rec = _density.get(key[0], {}).get(key[1:])
    if rec and rec.get("pending") is not None:
        pinned, ev, pixels = rec["pending"]
        if ev.query():
            rec["d"] = float(pinned[1]) / max(pixels, 1)
            rec["pending"] = None
    return rec


def _route_sparse(key):
    """the gradient's density is a property of the loss behind the layer: it is read back WITHOUT a sync (pinned copy + event, looked at
    on a later call) and a layer whose gradient turns out dense is sent to the library route, re-probed now and then"""
    rec = _observe(key)
    if rec is None or rec.get("d") is None or rec["d"] <= max_density:
        return True
    rec["dense_calls"] = rec.get("dense_calls", 0) + 1
    if rec["dense_calls"] >= reprobe_every:
        rec["dense_calls"] = 0
        return True
    return False


def applicable(x, conv):
    if not _applicable(x, conv):
        return False
    return _route_sparse((conv.weight, x.shape[2], x.shape[3]))


def conv_ok(conv):
    """a convolution these kernels cover: 3 x 3 / pad 1 or 1 x 1 / pad 0, stride 1, biased, fp32, channel counts in whole 64s"""
    return (conv.stride == (1, 1)
            and ((conv.kernel_size == (3, 3) and conv.padding == (1, 1)) or (conv.kernel_size == (1, 1) and conv.padding == (0, 0)))
            and conv.padding_mode == "zeros" and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is not None
            and conv.in_channels % 64 == 0 and conv.out_channels % 64 == 0 and conv.weight.dtype == torch.float32)


def _applicable(x, conv):
    return (enabled and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and conv_ok(conv) and x.shape[1] == conv.in_channels
            and x.shape[0] * x.shape[2] * x.shape[3] >= min_pixels and x.shape[2] > 1 and x.shape[3] > 1
            and x.is_contiguous(memory_format=torch.channels_last) and torch.is_grad_enabled())


class _ConvBiasActSparse(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, act, key, support=None):
        ctx.key = key
        if support is not None:
            # loss-driven step (utils/retinahead.py:pixel_support): act(conv(x) + b) on the listed pixels only, zero elsewhere -- the
            # backward below never reads a pixel off that list
            z = forward_on_pixels(x, w, b, act, support)
        else:
            z = F.conv2d(x, w, None, 1, w.shape[2] // 2)
            N, C, H, W = z.shape
            ones, zeros = _ones_zeros(z.device, C)
            _lib.check(_lib.lib().odpl_bias_act_forward(_lib.ptr(z), N * H * W, C, _lib.ptr(b), _lib.ptr(ones), _lib.ptr(zeros), act, _lib.ptr(z),
                                                        _lib.current_stream(z.device)), "odpl_bias_act_forward")
        ctx.save_for_backward(x, w, z)
        ctx.act = act
        return z

    @staticmethod
    def backward(ctx, gy):
        x, w, y = ctx.saved_tensors
        N, C, H, W = y.shape
        R = N * H * W
        dev = y.device
        L = _lib.lib()
        if gy.dtype != torch.float32 or not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.float().contiguous(memory_format=torch.channels_last)
        gz = torch.empty_like(y)
        gb = torch.empty(C, dtype=torch.float32, device=dev)
        active = torch.empty(R, dtype=torch.uint8, device=dev)
        need = L.odpl_bn_act_workspace_bytes(R, C) + 4 * C
        ws = _workspace(dev, need)
        st = _lib.current_stream(dev)
        _lib.check(L.odpl_bias_act_backward_flags(_lib.ptr(y), _lib.ptr(gy), R, C, ctx.act, _lib.ptr(gz), _lib.ptr(gb), _lib.ptr(active),
                                                  _lib.ptr(ws), ws.numel(), st), "odpl_bias_act_backward_flags")
        need_x, need_w = ctx.needs_input_grad[0], ctx.needs_input_grad[1]
        gx = torch.empty_like(x) if need_x else None
        gw = torch.empty_like(w) if need_w else None
        d = _lib.SparseConv3x3()
        d.x, d.gy, d.active, d.w = x.data_ptr(), gz.data_ptr(), active.data_ptr(), w.data_ptr()
        d.grad_x = gx.data_ptr() if gx is not None else None
        d.grad_w = gw.data_ptr() if gw is not None else None
        for k in range(4):
            d.w_stride[k] = w.stride(k)
        d.B, d.H, d.W, d.Cin, d.Cout, d.ksize = N, H, W, x.shape[1], C, w.shape[2]
        stats = torch.empty(2, dtype=torch.int32, device=dev)
        d.stats = stats.data_ptr()
        nbytes = L.odpl_sparse_conv3x3_backward_workspace_bytes(ctypes.byref(d))
        if nbytes == 0:
            raise _lib.OdplError("odpl_sparse_conv3x3_backward_workspace_bytes rejected the descriptor")
        key = (dev, st.value, "sc")         # per stream: partials written by one launch are read by the next on the SAME stream
        ws2 = _ws2.get(key)
        if ws2 is None or ws2.numel() < nbytes:
            ws2 = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=dev)
            _ws2[key] = ws2
        _lib.check(L.odpl_sparse_conv3x3_backward(ctypes.byref(d), _lib.ptr(ws2), ws2.numel(), st), "odpl_sparse_conv3x3_backward")
        rec = _density.setdefault(ctx.key[0], {}).setdefault(ctx.key[1:], {"d": None, "pending": None, "dense_calls": 0})
        if rec["pending"] is None and not torch.cuda.is_current_stream_capturing():       # one read-back in flight per layer, never waited for
            pinned = torch.empty(2, dtype=torch.int32, pin_memory=True)
            pinned.copy_(stats, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            rec["pending"] = (pinned, ev, R)
        return gx, gw, gb, None, None, None


def forward_on_pixels(x, w, b, act, support, out=None, out_strides=None):
    """`odpl_sparse_conv3x3_forward`: act(conv3x3(x) + b) at the pixels of support = (list, count) (device int32 tensors), written into
    `out` (default: a zeroed channels-last map) with element strides out_strides = (batch, pixel)."""
    N, Cin, H, W = x.shape
    C = w.shape[0]
    lst, cnt = support
    if not (lst.is_cuda and cnt.is_cuda and lst.dtype == torch.int32 and cnt.dtype == torch.int32 and lst.is_contiguous()
            and lst.numel() >= N * H * W and cnt.numel() >= 1):
        raise ValueError("support = (int32 device list with room for every pixel of the map, int32 device count)")
    if not (x.is_contiguous(memory_format=torch.channels_last) and x.dtype == torch.float32 and w.dtype == torch.float32 and w.shape[1] == Cin
            and w.shape[2] == w.shape[3] and w.shape[2] in (1, 3)):
        raise ValueError("forward_on_pixels needs a dense channels-last fp32 input and fp32 weights [Cout, Cin, k, k], k = 3 (pad 1) or 1")
    if out is None:
        out = torch.zeros((N, H, W, C), dtype=torch.float32, device=x.device).permute(0, 3, 1, 2)        # channels-last [N, C, H, W]
        out_strides = (H * W * C, C)
    d = _lib.SparseConv3x3Fwd()
    d.x, d.w, d.bias, d.y = x.data_ptr(), w.data_ptr(), (b.data_ptr() if b is not None else None), out.data_ptr()
    for k in range(4):
        d.w_stride[k] = w.stride(k)
    d.y_batch_stride, d.y_pixel_stride = out_strides
    d.list, d.count = support[0].data_ptr(), support[1].data_ptr()
    d.B, d.H, d.W, d.Cin, d.Cout, d.act, d.ksize = N, H, W, Cin, C, act, w.shape[2]
    st = _lib.current_stream(x.device)
    ws = _ws2.get((x.device, st.value, "fwd"))
    if ws is None:
        ws = torch.empty(32 << 20, dtype=torch.uint8, device=x.device)           # split-K partials of short lists (stream-ordered reuse:
        _ws2[(x.device, st.value, "fwd")] = ws                                         # one buffer per stream, two streams never share one)
    _lib.check(_lib.lib().odpl_sparse_conv3x3_forward(ctypes.byref(d), _lib.ptr(ws), ws.numel(), st), "odpl_sparse_conv3x3_forward")
    return out


def conv_bias_act(x, conv, act="relu", support=None):
    """`act(conv(x))` with the sparse backward; callers check `applicable(x, conv)` first.  support = (pixel list, count): evaluate
    the forward on those pixels only (a loss-driven training step)."""
    return _ConvBiasActSparse.apply(x, conv.weight, conv.bias, ACT[act], (conv.weight, x.shape[2], x.shape[3]), support)


class _UpAddOnPixels(torch.autograd.Function):
    """`F.interpolate(coarse, size=lateral.shape[2:], mode="bilinear", align_corners=False) + lateral` on the listed pixels of the fine
    map, zero elsewhere (the FPN's top-down merge in a loss-driven step).  backward: the gradient passes to `lateral` as it is and to
    `coarse` through ATen's bilinear backward."""

    @staticmethod
    def forward(ctx, coarse, lateral, support):
        N, C, H, W = lateral.shape
        h, w = coarse.shape[2], coarse.shape[3]
        lst, cnt = support
        if not (coarse.is_contiguous(memory_format=torch.channels_last) and lateral.is_contiguous(memory_format=torch.channels_last)
                and coarse.dtype == lateral.dtype == torch.float32 and coarse.shape[:2] == lateral.shape[:2] and C % 4 == 0
                and lst.dtype == torch.int32 and lst.numel() >= N * H * W):
            raise ValueError("up_add_on_pixels needs dense channels-last fp32 maps with equal batch and channels and an int32 pixel list")
        out = torch.zeros((N, H, W, C), dtype=torch.float32, device=lateral.device).permute(0, 3, 1, 2)
        _lib.check(_lib.lib().odpl_sparse_upsample_add(_lib.ptr(coarse), _lib.ptr(lateral), _lib.ptr(out), _lib.ptr(lst), _lib.ptr(cnt),
                                                       N, C, h, w, H, W, _lib.current_stream(lateral.device)), "odpl_sparse_upsample_add")
        ctx.sizes = ((H, W), tuple(coarse.shape))
        return out

    @staticmethod
    def backward(ctx, g):
        (H, W), cshape = ctx.sizes
        gc = None
        if ctx.needs_input_grad[0]:
            gc = torch.ops.aten.upsample_bilinear2d_backward(g.contiguous(memory_format=torch.channels_last), [H, W], list(cshape), False, None, None)
        return gc, (g if ctx.needs_input_grad[1] else None), None


def up_add_on_pixels(coarse, lateral, support):
    return _UpAddOnPixels.apply(coarse, lateral, support)


def union_lists(B, H, W, own=None, finer=None, finer_hw=None):
    """`odpl_pixel_union_lists`: (marked, near, counts) for an H x W map -- marked = pixels in `own` = (list, count) or bilinear sources
    of the pixels in `finer` = (list, count) of a finer_hw map; near = within one pixel of a marked one.  -> ((marked, count),
    (near, count)) as `forward_on_pixels` supports."""
    L = _lib.lib()
    if own is None and finer is None:
        raise _lib.OdplError("union_lists needs a list of this map's pixels, of a finer map's pixels, or both")
    dev = (own or finer)[0].device
    P = B * H * W
    marked = torch.empty(P, dtype=torch.int32, device=dev)
    near = torch.empty(P, dtype=torch.int32, device=dev)
    counts = torch.empty(2, dtype=torch.int32, device=dev)
    n = L.odpl_pixel_union_lists_workspace_bytes(B, H, W)
    if n == 0:
        raise _lib.OdplError("odpl_pixel_union_lists_workspace_bytes rejected the map")
    ws = torch.empty(n, dtype=torch.uint8, device=dev)
    Hf, Wf = finer_hw if finer is not None else (H, W)
    _lib.check(L.odpl_pixel_union_lists(own[0].data_ptr() if own else None, own[1].data_ptr() if own else None,
                                        finer[0].data_ptr() if finer else None, finer[1].data_ptr() if finer else None,
                                        B, H, W, Hf, Wf, _lib.ptr(marked), _lib.ptr(near), _lib.ptr(counts), _lib.ptr(ws), ws.numel(),
                                        _lib.current_stream(dev)), "odpl_pixel_union_lists")
    return (marked, counts[0:1]), (near, counts[1:2])
