"""YOLOv5 s/m/l/x as the CALLER of the hot path; mirrors `model/YOLOV5.py:20-218` + `utils/YoloV5Utils.py:37-127`:
class name, ctor `(classes, args)` with `args.type in {Yolov5s, Yolov5m, Yolov5l, Yolov5x}`, forward -> list of
[B, 3, G, G, 5+C] tensors (fine -> coarse), identical `state_dict` keys (`backbone_head.seq<k>_<Kind>...`).
BN + LeakyReLU epilogues run in the fused HIP kernels on channels-last fp32 activations."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..LightningFunc import lightningUtils as _lu
from ..LightningFunc import optimizer as _opt
from ..LightningFunc import step as _step
from ..LightningFunc.accuracy import get_yolo_statistics as _get_yolo_statistics
from ..LightningFunc.losses import configure_loss
from ..LightningFunc.nms import yolo_non_max_suppression
from ..LightningFunc.utils import bnact as _bnact
from ..LightningFunc.utils.headconv import HeadConv2d
from ..LightningFunc.utils.s2d import focus as _focus
from ..LightningFunc.utils.spp import spp_cat
from ..LightningFunc.utils.spp import tile_path as _spp_tile_path
from ..LightningFunc.utils.upsample import upsample_cat, upsample_nearest

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module

MULTIPLES = {"Yolov5s": (0.33, 0.5), "Yolov5m": (0.67, 0.75), "Yolov5l": (1.0, 1.0), "Yolov5x": (1.33, 1.25)}


def _w(c, wm):
    return int(round(c * wm, 1))                 # YoloV5Utils.py:41-44


def _bn_act(x, bn, act, residual=None):
    if _bnact.fusable(x, bn):
        return _bnact.bn_act(x, bn, "leaky" if act else "linear", residual)
    y = bn(x)
    y = F.leaky_relu(y, 0.1) if act else y
    return y if residual is None else residual + y


class Conv(nn.Module):
    """conv + BN + LeakyReLU(0.1); parameters under `conv`, `bn` (YoloV5Utils.py:37-53)."""

    def __init__(self, c1, c2, k=1, s=1, g=1, act=True, width_multiple=1.0):
        super().__init__()
        c1, c2 = _w(c1, width_multiple), _w(c2, width_multiple)
        self.conv = nn.Conv2d(c1, c2, k, s, k // 2, groups=g, bias=False)
        self.bn = nn.BatchNorm2d(c2)
        self.act = nn.LeakyReLU(0.1, inplace=True) if act else nn.Identity()
        self._has_act = bool(act)

    def forward(self, x, residual=None):
        return _bn_act(self.conv(x), self.bn, self._has_act, residual)


class Bottleneck(nn.Module):
    def __init__(self, c1, c2, shortcut=True, g=1, e=0.5, width_multiple=1.0):
        super().__init__()
        c1, c2 = _w(c1, width_multiple), _w(c2, width_multiple)
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_, c2, 3, 1, g=g)
        self.add = shortcut and c1 == c2

    def forward(self, x):
        return self.cv2(self.cv1(x), residual=x if self.add else None)


class BottleneckCSP(nn.Module):
    def __init__(self, c1, c2, n=1, shortcut=True, g=1, e=0.5, width_multiple=1.0, depth_multiple=1.0):
        super().__init__()
        c1, c2 = _w(c1, width_multiple), _w(c2, width_multiple)
        n = int(round(n * depth_multiple, 1))
        c_ = int(c2 * e)
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = nn.Conv2d(c1, c_, 1, 1, bias=False)
        self.cv3 = nn.Conv2d(c_, c_, 1, 1, bias=False)
        self.cv4 = Conv(c2, c2, 1, 1)
        self.bn = nn.BatchNorm2d(2 * c_)
        self.act = nn.LeakyReLU(0.1, inplace=True)
        self.m = nn.Sequential(*[Bottleneck(c_, c_, shortcut, g, e=1.0) for _ in range(n)])

    def forward(self, x):
        y1, y2 = self.cv3(self.m(self.cv1(x))), self.cv2(x)
        y = _bnact.bn_act_of_cat(y1, y2, self.bn, "leaky") if (y1.is_cuda and self.bn.momentum is not None) else None
        if y is None:
            y = _bn_act(torch.cat((y1, y2), 1), self.bn, True)
        return self.cv4(y)


class SPP(nn.Module):
    def __init__(self, c1, c2, k=(5, 9, 13), width_multiple=1.0):
        super().__init__()
        c1, c2 = _w(c1, width_multiple), _w(c2, width_multiple)
        c_ = c1 // 2
        self.cv1 = Conv(c1, c_, 1, 1)
        self.cv2 = Conv(c_ * (len(k) + 1), c2, 1, 1)
        self.k = tuple(k)
        self.m = nn.ModuleList([nn.MaxPool2d(kernel_size=x, stride=1, padding=x // 2) for x in k])

    def forward(self, x):
        x = self.cv1(x)
        if self.k == (5, 9, 13) and x.is_cuda and (_spp_tile_path(x) if x.dtype == torch.bfloat16 else
                                                  (x.dtype == torch.float32 and x.shape[2] * x.shape[3] <= 1024)):
            return self.cv2(spp_cat(x, "v5"))                             # fused HIP SPP, order [x, m5, m9, m13] (YoloV5Utils.py:110)
        return self.cv2(torch.cat([x] + [m(x) for m in self.m], 1))


class Focus(nn.Module):
    def __init__(self, c1, c2, k=1, width_multiple=1.0):
        super().__init__()
        self.conv = Conv(c1 * 4, _w(c2, width_multiple), k, 1)

    def forward(self, x):
        if x.is_cuda and x.dtype == torch.float32 and x.shape[2] % 2 == 0 and x.shape[3] % 2 == 0:
            cl = self.conv.conv.weight.is_contiguous(memory_format=torch.channels_last) and not self.conv.conv.weight.is_contiguous()
            return self.conv(_focus(x, cl or None))             # one HIP shuffle, written in the stem conv's memory format
        return self.conv(torch.cat([x[..., ::2, ::2], x[..., 1::2, ::2], x[..., ::2, 1::2], x[..., 1::2, 1::2]], 1))


class backbone_head(nn.Module):
    def __init__(self, num_classes, num_anchors, yolov5_type):
        super().__init__()
        dm, wm = MULTIPLES[yolov5_type]
        self.no = (num_classes + 5) * num_anchors
        kw, kd = dict(width_multiple=wm), dict(width_multiple=wm, depth_multiple=dm)
        self.seq0_Focus = Focus(3, 64, 3, **kw)
        self.seq1_Conv = Conv(64, 128, 3, 2, **kw)
        self.seq2_Bottleneck = Bottleneck(128, 128, **kw)
        self.seq3_Conv = Conv(128, 256, 3, 2, **kw)
        self.seq4_BottleneckCSP = BottleneckCSP(256, 256, 9, **kd)
        self.seq5_Conv = Conv(256, 512, 3, 2, **kw)
        self.seq6_BottleneckCSP = BottleneckCSP(512, 512, 9, **kd)
        self.seq7_Conv = Conv(512, 1024, 3, 2, **kw)
        self.seq8_SPP = SPP(1024, 1024, [5, 9, 13], **kw)
        self.seq9_BottleneckCSP = BottleneckCSP(1024, 1024, 6, **kd)
        self.seq10_BottleneckCSP = BottleneckCSP(1024, 1024, 3, False, **kd)
        self.seq11_Conv2d = HeadConv2d(_w(1024, wm), self.no, 1, 1, 0)        # Detect.m: NCHW heads + structured backward (utils/headconv.py)
        self.seq14_Conv = Conv(1536, 512, 1, 1, **kw)
        self.seq15_BottleneckCSP = BottleneckCSP(512, 512, 3, False, **kd)
        self.seq16_Conv2d = HeadConv2d(_w(512, wm), self.no, 1, 1, 0)
        self.seq19_Conv = Conv(768, 256, 1, 1, **kw)
        self.seq20_BottleneckCSP = BottleneckCSP(256, 256, 3, False, **kd)
        self.seq21_Conv2d = HeadConv2d(_w(256, wm), self.no, 1, 1, 0)

    @staticmethod
    def _up2(x):
        return upsample_nearest(x, (x.shape[2] * 2, x.shape[3] * 2))

    def forward(self, x):
        x = self.seq3_Conv(self.seq2_Bottleneck(self.seq1_Conv(self.seq0_Focus(x))))
        r0 = self.seq4_BottleneckCSP(x)
        r1 = self.seq6_BottleneckCSP(self.seq5_Conv(r0))
        x = self.seq9_BottleneckCSP(self.seq8_SPP(self.seq7_Conv(r1)))
        route = self.seq10_BottleneckCSP(x)
        head = lambda conv, t: conv(t)                                         # noqa: E731  (HeadConv2d on the device, plain conv on the CPU)
        out0 = head(self.seq11_Conv2d, route)
        route = self.seq15_BottleneckCSP(self.seq14_Conv(upsample_cat(route, r1)))
        out1 = head(self.seq16_Conv2d, route)
        x = self.seq20_BottleneckCSP(self.seq19_Conv(upsample_cat(route, r0)))
        return head(self.seq21_Conv2d, x), out1, out0


class Yolo_Layers(nn.Module):
    """YOLOV5.py:81-97: [B, na*no, G, G] -> [B, na, G, G, no].  The reference makes the permuted tensor contiguous (a full copy
    per head per step); here an NCHW-contiguous head (what `HeadConv2d` writes) is returned as the permuted VIEW -- same shape,
    same values, strides (na*no*G2, no*G2, G, 1, G2) -- which the criterion consumes in place (`ODPL_LAYOUT_PLANAR`) and the
    NMS makes contiguous itself when it needs the reference's memory reinterpretation (YOLOV5.py:176-182)."""

    def __init__(self, nc=80, anchors=()):
        super().__init__()
        self.no, self.nl, self.na = nc + 5, len(anchors), len(anchors[0]) // 2

    view_heads = True               # class switch: False = always .contiguous() like the reference

    def forward(self, x, inference=False):
        out = []
        for t in x:
            v = t.view(t.shape[0], self.na, self.no, t.shape[2], t.shape[3]).permute(0, 1, 3, 4, 2)
            if Yolo_Layers.view_heads and t.is_cuda and t.is_contiguous():
                if hasattr(t, "_odpl_head_src"):
                    v._odpl_head_base = t                     # lets the criterion run the Detect convolution's backward itself
                out.append(v)
            else:
                out.append(v.contiguous())
        return out


class YOLOv5(_Base):
    anchors = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
    anch_masks = None
    img_size = 640
    ignore_thres = 0.5
    inference = False

    def __init__(self, classes, args):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "YOLOv5"
        self.num_anchors = len(self.anchors[0]) // 2
        self.backbone_head = backbone_head(self.num_classes, self.num_anchors, args.type)
        self.yolo_layers = Yolo_Layers(self.num_classes, anchors=self.anchors)
        self.sample = (2, 3, self.img_size, self.img_size)
        self.criterion = configure_loss(args, None, self.anchors, None, self.num_classes, self.img_size)
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)

    def forward(self, x):
        return self.yolo_layers(list(self.backbone_head(x)), self.inference)

    def non_max_suppression(self, predictions, conf_thres=0.5, nms_thres=0.4):
        return yolo_non_max_suppression(predictions, nms_thres=nms_thres)

    training_step = _step.training_step
    training_epoch_end = _step.training_epoch_end
    validation_step = _step.validation_step
    validation_epoch_end = _step.validation_epoch_end
    test_step = _step.test_step
    test_epoch_end = _step.test_epoch_end
    write_Best_model_path = _lu.write_Best_model_path
    read_Best_model_path = _lu.read_Best_model_path
    configure_optimizers = _opt.configure_optimizers
    get_yolo_statistics = _get_yolo_statistics
