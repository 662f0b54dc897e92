"""Fused BatchNorm2d + activation epilogue of the reference's Conv_Bn_Activation unit
(`utils/YoloV4Utils.py:208-236`) for channels-last fp32 activations: 2 passes forward, 2 backward."""
import torch

from ... import _lib

ACT = {"linear": 0, "mish": 1, "leaky": 2, "relu": 3}
_ws = {}


def _workspace(dev, need):
    w = _ws.get(dev)
    if w is None or w.numel() < need:
        w = torch.empty(int(need * 1.5) + 4096, dtype=torch.uint8, device=dev)
        _ws[dev] = w
    return w


def fusable(x, bn):
    """Channels-last fp32 activations, or bf16 ones (a bf16-autocast network) when the C++ binding is built; fp32 affine BN."""
    ok_dtype = x.dtype == torch.float32 or (x.dtype == torch.bfloat16 and _lib.torch_ext() is not None)
    return (x.is_cuda and ok_dtype and x.dim() == 4 and x.shape[1] % 4 == 0 and bn.momentum is not None
            and bn.affine and bn.track_running_stats and bn.weight.dtype == torch.float32
            and x.is_contiguous(memory_format=torch.channels_last))


class _BnActFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, running_mean, running_var, training, momentum, eps, act, nbt=None):
        N, C, H, W = x.shape
        R = N * H * W
        dev = x.device
        L = _lib.lib()
        y = torch.empty_like(x)                                  # keeps the channels-last strides
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean = running_mean.clone()
            invstd = torch.rsqrt(running_var + eps)
        need = L.odpl_bn_act_workspace_bytes(R, C)
        ws = _workspace(dev, need)
        _lib.check(L.odpl_bn_act_forward(_lib.ptr(x), R, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean), _lib.ptr(running_var),
                                         1 if training else 0, float(momentum), float(eps), act, _lib.ptr(mean), _lib.ptr(invstd),
                                         _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.ptr(nbt) if training else None,
                                         _lib.current_stream(dev)), "odpl_bn_act_forward")
        ctx.save_for_backward(x, weight, bias, mean, invstd)
        ctx.training, ctx.act = training, act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, invstd = ctx.saved_tensors
        N, C, H, W = x.shape
        R = N * H * W
        dev = x.device
        L = _lib.lib()
        if not gy.is_contiguous(memory_format=torch.channels_last) or gy.dtype != torch.float32:
            gy = gy.float().contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        need = L.odpl_bn_act_workspace_bytes(R, C)
        ws = _workspace(dev, need)
        _lib.check(L.odpl_bn_act_backward(_lib.ptr(x), _lib.ptr(gy), R, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(mean), _lib.ptr(invstd),
                                          1 if ctx.training else 0, ctx.act, _lib.ptr(gx), _lib.ptr(dg), _lib.ptr(db), _lib.ptr(ws),
                                          ws.numel(), _lib.current_stream(dev)), "odpl_bn_act_backward")
        return gx, dg, db, None, None, None, None, None, None, None


def bn_act(x, bn, act, residual=None):
    """act(bn(x)) [+ residual] for a channels-last fp32 tensor and an nn.BatchNorm2d module (its buffers are updated like
    nn.BatchNorm2d does in training mode).  Uses the C++ binding when built (10x less host time per call)."""
    ext = _lib.torch_ext()
    if ext is not None:
        y = ext.bn_act(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training,
                       bn.momentum, bn.eps, ACT[act], residual)
        if y is not None:
            return y
    if residual is not None:
        return residual + bn_act(x, bn, act)
    return _BnActFn.apply(x, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, ACT[act],
                          bn.num_batches_tracked)


class _BiasAddFn(torch.autograd.Function):
    """z + bias[None, :, None, None] in place; backward sums the (NCHW-dense) output gradient per channel with one HIP
    launch instead of ATen's `sum((0, 2, 3))` (0.2-0.5 ms on the head shapes)."""

    @staticmethod
    def forward(ctx, z, bias):
        ctx.mark_dirty(z)
        z.add_(bias.view(1, -1, 1, 1))
        return z

    @staticmethod
    def backward(ctx, gy):
        if gy.is_cuda and gy.dtype in (torch.float32, torch.bfloat16) and gy.is_contiguous():
            B, C, H, W = gy.shape
            gb = torch.empty(C, dtype=torch.float32, device=gy.device)
            _lib.check(_lib.lib().odpl_bias_grad_nchw(_lib.ptr(gy), 1 if gy.dtype == torch.bfloat16 else 0, B, C, H * W, _lib.ptr(gb),
                                                      _lib.current_stream(gy.device)), "odpl_bias_grad_nchw")
        else:
            gb = gy.sum((0, 2, 3))
        return gy, gb


def conv_bias(x, conv):
    """`conv(x)` for an nn.Conv2d with a bias (the linear head convs, `model/YOLOV4.py:89-127`): bias-free convolution on
    MIOpen, bias added in place, bias gradient by the HIP channel-sum kernel."""
    z = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return _BiasAddFn.apply(z, conv.bias)          # under bf16 autocast z is bf16, the bias and its gradient stay fp32


class _BnAddActFn(torch.autograd.Function):
    """y = act(bn(x) + s) in one write; backward: gu = gy * act'(y) (also the gradient of s), then the BN backward on gu."""

    @staticmethod
    def forward(ctx, x, s, weight, bias, running_mean, running_var, training, momentum, eps, act, nbt):
        N, C, H, W = x.shape
        R = N * H * W
        dev = x.device
        L = _lib.lib()
        y = torch.empty_like(x)
        if training:
            mean = torch.empty(C, dtype=torch.float32, device=dev)
            invstd = torch.empty(C, dtype=torch.float32, device=dev)
        else:
            mean = running_mean.clone()
            invstd = torch.rsqrt(running_var + eps)
        need = L.odpl_bn_act_workspace_bytes(R, C)
        ws = _workspace(dev, need)
        _lib.check(L.odpl_bn_add_act_forward(_lib.ptr(x), _lib.ptr(s), R, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(running_mean),
                                             _lib.ptr(running_var), 1 if training else 0, float(momentum), float(eps), act, _lib.ptr(mean),
                                             _lib.ptr(invstd), _lib.ptr(y), _lib.ptr(ws), ws.numel(), _lib.ptr(nbt) if training else None,
                                             _lib.current_stream(dev)), "odpl_bn_add_act_forward")
        ctx.save_for_backward(x, weight, bias, mean, invstd, y)
        ctx.training, ctx.act = training, act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, weight, bias, mean, invstd, y = ctx.saved_tensors
        N, C, H, W = x.shape
        R = N * H * W
        dev = x.device
        L = _lib.lib()
        if gy.dtype != torch.float32 or not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.float().contiguous(memory_format=torch.channels_last)
        gu = torch.empty_like(x)                                   # d loss / d s = gy * act'(y): written by the reduce pass, read by the dx pass
        gx = torch.empty_like(x)
        dg = torch.empty(C, dtype=torch.float32, device=dev)
        db = torch.empty(C, dtype=torch.float32, device=dev)
        need = L.odpl_bn_act_workspace_bytes(R, C)
        ws = _workspace(dev, need)
        _lib.check(L.odpl_bn_add_act_backward(_lib.ptr(x), _lib.ptr(y), _lib.ptr(gy), R, C, _lib.ptr(weight), _lib.ptr(bias), _lib.ptr(mean),
                                              _lib.ptr(invstd), 1 if ctx.training else 0, ctx.act, _lib.ptr(gx), _lib.ptr(gu), _lib.ptr(dg),
                                              _lib.ptr(db), _lib.ptr(ws), ws.numel(), _lib.current_stream(dev)), "odpl_bn_add_act_backward")
        return gx, gu, dg, db, None, None, None, None, None, None, None


def bn_add_act(x, bn, s, act="relu"):
    """act(bn(x) + s): the tail of a ResNet bottleneck (`model/backbone/RetinaNetbone.py`), fp32 channels-last on the device
    (mish is not supported: its derivative is not a function of the output)."""
    if not (fusable(x, bn) and x.dtype == torch.float32 and s.shape == x.shape and s.dtype == torch.float32 and act in ("relu", "leaky", "linear")
            and s.is_contiguous(memory_format=torch.channels_last) and not (s.is_contiguous() and x.shape[1] > 1 and x.shape[2] * x.shape[3] > 1)):
        y = bn(x) + s
        return torch.relu(y) if act == "relu" else (torch.nn.functional.leaky_relu(y, 0.1) if act == "leaky" else y)
    return _BnAddActFn.apply(x, s, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.training, bn.momentum, bn.eps, ACT[act],
                             bn.num_batches_tracked)


_consts = {}


def _ones_zeros(dev, C):
    k = (dev, C)
    v = _consts.get(k)
    if v is None:
        v = (torch.ones(C, dtype=torch.float32, device=dev), torch.zeros(C, dtype=torch.float32, device=dev))
        _consts[k] = v
    return v


class _BiasActFn(torch.autograd.Function):
    """y = act(z + bias) written over z (the bias-free convolution's output, which no backward needs); backward from the saved y."""

    @staticmethod
    def forward(ctx, z, bias, act):
        N, C, H, W = z.shape
        ones, zeros = _ones_zeros(z.device, C)
        _lib.check(_lib.lib().odpl_bias_act_forward(_lib.ptr(z), N * H * W, C, _lib.ptr(bias), _lib.ptr(ones), _lib.ptr(zeros), act, _lib.ptr(z),
                                                    _lib.current_stream(z.device)), "odpl_bias_act_forward")
        ctx.mark_dirty(z)
        ctx.save_for_backward(z)
        ctx.act = act
        return z

    @staticmethod
    def backward(ctx, gy):
        (y,) = ctx.saved_tensors
        N, C, H, W = y.shape
        R = N * H * W
        L = _lib.lib()
        if gy.dtype != torch.float32 or not gy.is_contiguous(memory_format=torch.channels_last):
            gy = gy.float().contiguous(memory_format=torch.channels_last)
        gx = torch.empty_like(y)
        gb = torch.empty(C, dtype=torch.float32, device=y.device)
        need = L.odpl_bn_act_workspace_bytes(R, C) + 4 * C
        ws = _workspace(y.device, need)
        _lib.check(L.odpl_bias_act_backward(_lib.ptr(y), _lib.ptr(gy), R, C, ctx.act, _lib.ptr(gx), _lib.ptr(gb), _lib.ptr(ws), ws.numel(),
                                            _lib.current_stream(y.device)), "odpl_bias_act_backward")
        return gx, gb, None


def conv_bias_act(x, conv, act="relu"):
    """`act(conv(x))` for an nn.Conv2d WITH a bias and no BatchNorm (RetinaNet's head units, `model/RetinaNet.py:93-99`): the
    bias-free convolution on MIOpen, bias + activation as one in-place epilogue, ReLU's backward and the bias gradient as one
    pass.  act: "relu" | "leaky" | "linear".  Falls back to the torch ops off the device / off channels-last fp32."""
    z = torch.nn.functional.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    ok = (z.is_cuda and z.dtype == torch.float32 and z.dim() == 4 and z.shape[1] % 4 == 0 and conv.bias is not None
          and conv.bias.dtype == torch.float32 and z.is_contiguous(memory_format=torch.channels_last) and not z.is_contiguous())
    if not ok:
        if conv.bias is not None:
            z = z + conv.bias.view(1, -1, 1, 1)
        return torch.relu(z) if act == "relu" else (torch.nn.functional.leaky_relu(z, 0.1) if act == "leaky" else z)
    return _BiasActFn.apply(z, conv.bias, ACT[act])


def bn_act_cat(x1, bn1, act1, x2, bn2, act2):
    """`torch.cat((act1(bn1(x1)), act2(bn2(x2))), 1)` with both epilogues writing their column slice of the result (no cat
    copy forward, no slice copies backward).  Returns None when the C++ binding is absent or the layout is not fusable."""
    ext = _lib.torch_ext()
    if ext is None or bn1.training != bn2.training:
        return None
    return ext.bn_act_cat(x1, bn1.weight, bn1.bias, bn1.running_mean, bn1.running_var, bn1.num_batches_tracked, bn1.momentum, bn1.eps, ACT[act1],
                          x2, bn2.weight, bn2.bias, bn2.running_mean, bn2.running_var, bn2.num_batches_tracked, bn2.momentum, bn2.eps, ACT[act2],
                          bn1.training)


bn_of_cat_enabled = True        # A/B switch (tools/bench_v5.py --no-bn-of-cat): False = torch.cat followed by the BN epilogue


def bn_act_of_cat(x1, x2, bn, act):
    """`act(bn(torch.cat((x1, x2), 1)))` for ONE nn.BatchNorm2d over the concatenated channels (YOLOv5's BottleneckCSP,
    `utils/YoloV5Utils.py:76-93`: `self.act(self.bn(torch.cat((y1, y2), dim=1)))`) without the cat: BatchNorm is per channel, so the
    two halves are two independent epilogues over slices of the module's parameters and buffers, each writing its column slice of the
    result; backward hands each convolution a dense gradient.  None when the C++ binding is absent or the layout is not fusable."""
    ext = _lib.torch_ext()
    c1 = x1.shape[1]
    if not bn_of_cat_enabled or ext is None or c1 % 4 or x2.shape[1] % 4 or bn.num_features != c1 + x2.shape[1] or not bn.affine or not bn.track_running_stats:
        return None
    return ext.bn_act_of_cat(x1, x2, bn.weight, bn.bias, bn.running_mean, bn.running_var, bn.num_batches_tracked, bn.training,
                             bn.momentum, bn.eps, ACT[act])

