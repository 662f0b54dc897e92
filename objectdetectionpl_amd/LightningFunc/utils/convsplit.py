"""Convolution whose WEIGHT-gradient kernel runs on a side HIP stream.

In the backward of a conv+BN+activation network the chain `BN backward -> conv backward-data -> BN backward -> ...` is the
critical path; the weight gradients (`wrw`) hang off it and nothing reads them until the optimiser (or the gradient exchange).
The BN passes are HBM-bound, the convolutions MFMA-bound: issuing `wrw` on a second stream lets the two kinds of kernels share
the chip instead of taking turns.  The math is untouched -- both halves are the same `aten::convolution_backward` MIOpen calls,
just with split `output_mask`s -- so gradients are bit-identical to the single-stream path for deterministic solvers.

Synchronisation: the side stream waits for the incoming gradient; every tensor that crosses streams is `record_stream`ed;
`join()` makes the current stream wait for all outstanding weight gradients and is called (a) from an end-of-backward callback
queued by the first split convolution of each backward pass and (b) by `parallel.GradAverager` before it packs a bucket."""
import torch

_side = {}
_pending = {}
enabled = False


def side_stream(device):
    s = _side.get(device)
    if s is None:
        s = torch.cuda.Stream(device=device)
        _side[device] = s
    return s


def join(device=None):
    """The current stream waits for the weight-gradient kernels issued so far (no host blocking)."""
    for dev, s in _side.items():
        if device is None or dev == device:
            torch.cuda.current_stream(dev).wait_stream(s)
            _pending[dev] = False


def _end_of_backward():
    join()


class _ConvSplit(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, stride, padding, dilation, groups):
        ctx.conf = (stride, padding, dilation, groups)
        ctx.save_for_backward(x, w)
        return torch.ops.aten.convolution(x, w, None, stride, padding, dilation, False, [0, 0], groups)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, padding, dilation, groups = ctx.conf
        dev = gy.device
        main, side = torch.cuda.current_stream(dev), side_stream(dev)
        gx = gw = None
        if ctx.needs_input_grad[1]:
            if not _pending.get(dev):
                _pending[dev] = True
                torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            side.wait_stream(main)                              # gy (and everything before it) is ready
            with torch.cuda.stream(side):
                gw = torch.ops.aten.convolution_backward(gy, x, w, None, stride, padding, dilation, False, [0, 0], groups,
                                                         [False, True, False])[1]
            for t in (gy, x, w):
                t.record_stream(side)                           # keep their memory from being reused under the side stream
            gw.record_stream(main)
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(gy, x, w, None, stride, padding, dilation, False, [0, 0], groups,
                                                     [True, False, False])[0]
        return gx, gw, None, None, None, None


def conv2d(x, conv):
    """`conv(x)` for a bias-free nn.Conv2d; weight gradient on the side stream when `convsplit.enabled` and training on the GPU."""
    if enabled and x.is_cuda and conv.bias is None and conv.weight.requires_grad and torch.is_grad_enabled() and conv.padding_mode == "zeros":
        return _ConvSplit.apply(x, conv.weight, list(conv.stride), list(conv.padding), list(conv.dilation), conv.groups)
    return conv(x)
