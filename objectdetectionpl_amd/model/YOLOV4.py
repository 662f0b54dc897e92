"""YOLOv4 (CSPDarknet53 + SPP/PAN neck + 3 heads) as the CALLER of the hot path.

Mirrors `model/YOLOV4.py:136-283` of the reference: same class name, ctor `(classes, args)`, class
attributes (`anchors, anch_masks, img_size, ignore_thres`), `forward(x) -> [x52, x26, x13]` raw head
tensors, `.criterion` from `configure_loss`, and the same `state_dict` keys/shapes (module tree
`down1..down5 / neek / head`, each conv unit a ModuleList named `conv`) so reference checkpoints load.
The convolutions are plain `nn.Conv2d` (MIOpen / MFMA) -- they are not the hand-written part; the
detector math behind `.criterion` and `.non_max_suppression` is (libodpl_hip.so).
The network is described by tables below and assembled by loops; nothing here is device-specific
(the reference hard-codes `.cuda()` in the ctor, `YOLOV4.py:151`).
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..LightningFunc import lightningUtils as _lu
from ..LightningFunc import optimizer as _opt
from ..LightningFunc import step as _step
from ..LightningFunc.accuracy import get_yolo_statistics as _get_yolo_statistics
from ..LightningFunc.losses import configure_loss
from ..LightningFunc.utils import bnact as _bnact
from ..LightningFunc.utils import convsplit as _cs
from ..LightningFunc.utils.headconv import HeadConv2d
from ..LightningFunc.utils.spp import spp_cat
from ..LightningFunc.utils.upsample import upsample_cat, upsample_nearest

try:                                    # PL is optional: the reference derives from pl.LightningModule
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:                       # noqa: BLE001 - PL absent in this image
    _Base = nn.Module


class ConvUnit(nn.Module):
    """conv [+ BN] [+ activation]; parameters live under `conv.0` / `conv.1` like the reference's unit."""

    def __init__(self, cin, cout, k, stride, act, bn=True, bias=False):
        super().__init__()
        head = (not bn) and act == "linear" and k == 1 and stride == 1
        layers = [(HeadConv2d if head else nn.Conv2d)(cin, cout, k, stride, (k - 1) // 2, bias=bias)]
        if bn:
            layers.append(nn.BatchNorm2d(cout))
        if act == "mish":
            layers.append(nn.Mish())
        elif act == "leaky":
            layers.append(nn.LeakyReLU(0.1, inplace=True))
        elif act != "linear":
            raise ValueError(act)
        self.conv = nn.ModuleList(layers)
        self._act = act if bn else None
        self._plain_bias = (not bn) and bias and act == "linear"

    def forward(self, x, residual=None):
        """act(bn(conv(x))) [+ residual]: the skip add of a ResBlock rides on the epilogue's write."""
        if residual is not None:
            if self._act is not None and ConvUnit.fuse_epilogue:
                z = _cs.conv2d(x, self.conv[0])
                if _bnact.fusable(z, self.conv[1]):
                    return _bnact.bn_act(z, self.conv[1], self._act, residual)
                for l in list(self.conv)[1:]:
                    z = l(z)
                return residual + z
            return residual + self.forward(x)
        if self._plain_bias and ConvUnit.fuse_epilogue and x.is_cuda and x.dtype == torch.float32 and not isinstance(self.conv[0], HeadConv2d):
            return _bnact.conv_bias(x, self.conv[0])                 # a biased linear conv that is not a head: bias gradient by the HIP channel sum
        if self._act is not None and ConvUnit.fuse_epilogue:
            z = _cs.conv2d(x, self.conv[0])
            if _bnact.fusable(z, self.conv[1]):
                return _bnact.bn_act(z, self.conv[1], self._act)     # fused BN + activation (HIP, channels-last)
            x = z
            rest = list(self.conv)[1:]
        else:
            rest = self.conv
        for l in rest:
            x = l(x)
        return x

    fuse_epilogue = True          # class switch: False = plain nn.BatchNorm2d + nn.Mish / nn.LeakyReLU


def cat_units(u1, x1, u2, x2):
    """torch.cat([u1(x1), u2(x2)], 1) for two conv+BN+activation units: their epilogues write straight into the merged tensor."""
    if ConvUnit.fuse_epilogue and u1._act is not None and u2._act is not None:
        z1, z2 = _cs.conv2d(x1, u1.conv[0]), _cs.conv2d(x2, u2.conv[0])
        if _bnact.fusable(z1, u1.conv[1]) and _bnact.fusable(z2, u2.conv[1]):
            y = _bnact.bn_act_cat(z1, u1.conv[1], u1._act, z2, u2.conv[1], u2._act)
            if y is not None:
                return y
        for l in list(u1.conv)[1:]:
            z1 = l(z1)
        for l in list(u2.conv)[1:]:
            z2 = l(z2)
        return torch.cat([z1, z2], 1)
    return torch.cat([u1(x1), u2(x2)], 1)


class ResChain(nn.Module):
    """n x (1x1 mish -> 3x3 mish, + skip); parameters under `module_list.<i>.<0|1>`."""

    def __init__(self, ch, n):
        super().__init__()
        self.module_list = nn.ModuleList(
            nn.ModuleList([ConvUnit(ch, ch, 1, 1, "mish"), ConvUnit(ch, ch, 3, 1, "mish")]) for _ in range(n))

    def forward(self, x):
        for a, b in self.module_list:
            x = b(a(x), residual=x)
        return x


class CSPFirst(nn.Module):
    """Stage 1 (no ResChain module in the reference's layout: conv1..conv8)."""

    def __init__(self):
        super().__init__()
        spec = [(3, 32, 3, 1), (32, 64, 3, 2), (64, 64, 1, 1), (64, 64, 1, 1),
                (64, 32, 1, 1), (32, 64, 3, 1), (64, 64, 1, 1), (128, 64, 1, 1)]
        for i, (ci, co, k, s) in enumerate(spec, 1):
            setattr(self, f"conv{i}", ConvUnit(ci, co, k, s, "mish"))

    def forward(self, x):
        x2 = self.conv2(self.conv1(x))
        x4 = self.conv4(x2)
        x6 = self.conv6(self.conv5(x4), residual=x4)
        return self.conv8(cat_units(self.conv7, x6, self.conv3, x2))          # cat([conv7(x6), conv3(x2)], 1)


class CSPStage(nn.Module):
    """Stages 2-5: stride-2 conv, split, ResChain on one branch, merge."""

    def __init__(self, cin, cout, nblocks, half):
        super().__init__()
        self.conv1 = ConvUnit(cin, cout, 3, 2, "mish")
        self.conv2 = ConvUnit(cout, half, 1, 1, "mish")
        self.conv3 = ConvUnit(cout, half, 1, 1, "mish")
        self.resblock = ResChain(half, nblocks)
        self.conv4 = ConvUnit(half, half, 1, 1, "mish")
        self.conv5 = ConvUnit(2 * half, cout, 1, 1, "mish")

    def forward(self, x):
        x1 = self.conv1(x)
        return self.conv5(cat_units(self.conv4, self.resblock(self.conv3(x1)), self.conv2, x1))   # cat([conv4(...), conv2(x1)], 1)


class PANNeck(nn.Module):
    _spec = [(1024, 512, 1), (512, 1024, 3), (1024, 512, 1), (2048, 512, 1), (512, 1024, 3), (1024, 512, 1),
             (512, 256, 1), (512, 256, 1), (512, 256, 1), (256, 512, 3), (512, 256, 1), (256, 512, 3),
             (512, 256, 1), (256, 128, 1), (256, 128, 1), (256, 128, 1), (128, 256, 3), (256, 128, 1),
             (128, 256, 3), (256, 128, 1)]

    def __init__(self):
        super().__init__()
        for i, (ci, co, k) in enumerate(self._spec, 1):
            setattr(self, f"conv{i}", ConvUnit(ci, co, k, 1, "leaky"))

    def _seq(self, x, lo, hi):
        for i in range(lo, hi + 1):
            x = getattr(self, f"conv{i}")(x)
        return x

    @staticmethod
    def _merge(lateral, small):
        """torch.cat([lateral, Upsample(small)], 1) (YoloV4Utils.py:444-462); one buffer when `small` is exactly half the size."""
        if small.shape[2] * 2 == lateral.shape[2] and small.shape[3] * 2 == lateral.shape[3]:
            return upsample_cat(small, lateral, up_first=False)
        return torch.cat([lateral, upsample_nearest(small, lateral.shape[2:])], 1)

    def forward(self, d5, d4, d3):
        x3 = self._seq(d5, 1, 3)
        if x3.is_cuda and x3.dtype == torch.float32 and x3.shape[2] * x3.shape[3] <= 1024:
            spp = spp_cat(x3)                      # fused HIP SPP (YoloV4Utils.py:412-418)
        else:
            spp = torch.cat([F.max_pool2d(x3, 13, 1, 6), F.max_pool2d(x3, 9, 1, 4), F.max_pool2d(x3, 5, 1, 2), x3], 1)
        x6 = self._seq(spp, 4, 6)
        x13 = self._seq(self._merge(self.conv8(d4), self.conv7(x6)), 9, 13)
        x20 = self._seq(self._merge(self.conv15(d3), self.conv14(x13)), 16, 20)
        return x20, x13, x6


class Yolov4Head(nn.Module):
    def __init__(self, output_ch, num_classes, inference=False):
        super().__init__()
        self.inference = inference
        self.num_classes = num_classes
        L, N = "leaky", "linear"
        spec = {1: (128, 256, 3, 1, L), 2: (256, output_ch, 1, 1, N), 3: (128, 256, 3, 2, L),
                4: (512, 256, 1, 1, L), 5: (256, 512, 3, 1, L), 6: (512, 256, 1, 1, L), 7: (256, 512, 3, 1, L),
                8: (512, 256, 1, 1, L), 9: (256, 512, 3, 1, L), 10: (512, output_ch, 1, 1, N),
                11: (256, 512, 3, 2, L), 12: (1024, 512, 1, 1, L), 13: (512, 1024, 3, 1, L), 14: (1024, 512, 1, 1, L),
                15: (512, 1024, 3, 1, L), 16: (1024, 512, 1, 1, L), 17: (512, 1024, 3, 1, L),
                18: (1024, output_ch, 1, 1, N)}
        for i, (ci, co, k, s, act) in spec.items():
            lin = act == N
            setattr(self, f"conv{i}", ConvUnit(ci, co, k, s, act, bn=not lin, bias=lin))

    def _seq(self, x, lo, hi):
        for i in range(lo, hi + 1):
            x = getattr(self, f"conv{i}")(x)
        return x

    def forward(self, in52, in26, in13):
        x2 = self.conv2(self.conv1(in52))
        x8 = self._seq(torch.cat([self.conv3(in52), in26], 1), 4, 8)
        x10 = self.conv10(self.conv9(x8))
        x18 = self._seq(torch.cat([self.conv11(x8), in13], 1), 12, 18)
        return [x2, x10, x18]


class YOLOv4(_Base):
    anchors = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
    anch_masks = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
    img_size = 416
    grid_size = 0
    ignore_thres = 0.5

    def __init__(self, classes, args):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "YOLOv4"
        out_ch = (4 + 1 + self.num_classes) * 3
        self.down1 = CSPFirst()
        self.down2 = CSPStage(64, 128, 2, 64)
        self.down3 = CSPStage(128, 256, 8, 128)
        self.down4 = CSPStage(256, 512, 8, 256)
        self.down5 = CSPStage(512, 1024, 4, 512)
        self.neek = PANNeck()                      # sic: the reference's attribute name (state_dict keys)
        self.head = Yolov4Head(out_ch, self.num_classes, False)
        self.sample = (1, 3, self.img_size, self.img_size)
        self.criterion = configure_loss(args, None, self.anchors, self.anch_masks, self.num_classes, self.img_size)
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)

    @property
    def sampleImg(self):
        p = next(self.parameters())
        return torch.rand(self.sample, device=p.device)

    def forward(self, input):
        d1 = self.down1(input)
        d2 = self.down2(d1)
        d3 = self.down3(d2)
        d4 = self.down4(d3)
        d5 = self.down5(d4)
        x20, x13, x6 = self.neek(d5, d4, d3)
        return self.head(x20, x13, x6)

    # step glue: the functions the reference setattr's in __build_func (YOLOV4.py:186-202)
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

    def non_max_suppression(self, predictions, conf_thres=0.5, nms_thres=0.4):
        from ..LightningFunc.nms import yolo_non_max_suppression
        return yolo_non_max_suppression(predictions, nms_thres=nms_thres)
