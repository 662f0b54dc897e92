"""YOLOv3 (Darknet-53, lightnet naming) as the CALLER of the hot path; mirrors `model/YOLOV3.py:18-335` and the layer
helpers of `utils/YoloV3Utils.py`: class name, ctor `(classes, args)`, anchors divided by the stride in the ctor
(YOLOV3.py:54-56), `forward(x) -> [x13, x26, x52]` (coarse -> fine), identical `state_dict` keys
(`extractor.<n>_convbatch.layers.0/1`, `extractor.<letter>_residual.<n>_convbatch...`, `detector.<i>...`).
Darknet-53 is generated from a stage table; BN (momentum 0.01) + LeakyReLU epilogues use the fused HIP kernels."""
import os
from collections import OrderedDict

import torch
import torch.nn as nn

from ..LightningFunc import lightningUtils as _lu
from ..LightningFunc import optimizer as _opt
from ..LightningFunc import step as _step
from ..LightningFunc.accuracy import get_yolo_statistics as _get_yolo_statistics
from ..LightningFunc.losses import configure_loss
from ..LightningFunc.nms import yolo_non_max_suppression
from ..LightningFunc.utils import bnact as _bnact
from ..LightningFunc.utils.headconv import HeadConv2d
from ..LightningFunc.utils.upsample import upsample_nearest

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module


class _Upsample2x(nn.Upsample):
    """nn.Upsample(scale_factor=2, mode="nearest") (`model/YOLOV3.py:119-121` of the reference); channels-last device tensors take
    the HIP copy kernels of utils/upsample.py."""

    def forward(self, x):
        return upsample_nearest(x, (x.shape[2] * 2, x.shape[3] * 2))


class Conv2dBatchReLU(nn.Module):
    """conv + BN(momentum 0.01) + LeakyReLU(0.1); parameters under `layers.0`, `layers.1` (YoloV3Utils.py:154-197)."""

    def __init__(self, cin, cout, k, stride, pad, momentum=0.01):
        super().__init__()
        self.layers = nn.Sequential(nn.Conv2d(cin, cout, k, stride, pad, bias=False), nn.BatchNorm2d(cout, momentum=momentum),
                                    nn.LeakyReLU(0.1))

    def forward(self, x, residual=None):
        z = self.layers[0](x)
        if _bnact.fusable(z, self.layers[1]):
            return _bnact.bn_act(z, self.layers[1], "leaky", residual)      # the skip add rides on the epilogue's write
        z = self.layers[2](self.layers[1](z))
        return z if residual is None else residual + z


class Residual(nn.Sequential):
    def forward(self, x):
        mods = list(self)
        y = x
        for m in mods[:-1]:
            y = m(y)
        return mods[-1](y, residual=x) if isinstance(mods[-1], Conv2dBatchReLU) else x + mods[-1](y)


class SelectiveSequential(nn.Sequential):
    """Sequential that also returns the outputs of the named layers (YoloV3Utils.py:8-44)."""

    def __init__(self, selection, layers):
        super().__init__(layers)
        self.selection = [str(s) for s in selection]
        self.flatten = len(self.selection) == 1

    def forward(self, x):
        sel = {}
        for key, module in self._modules.items():
            x = module(x)
            if key in self.selection:
                sel[key] = x
        return (x, sel[self.selection[0]]) if self.flatten else (x, sel)


def _darknet53():
    """1_convbatch .. 52_convbatch with residual blocks a..w: stages of (channels, number of residuals)."""
    layers = OrderedDict()
    n, letter = 1, ord("a")
    layers[f"{n}_convbatch"] = Conv2dBatchReLU(3, 32, 3, 1, 1); n += 1
    cin = 32
    for cout, reps in ((64, 1), (128, 2), (256, 8), (512, 8), (1024, 4)):
        layers[f"{n}_convbatch"] = Conv2dBatchReLU(cin, cout, 3, 2, 1); n += 1
        for _ in range(reps):
            blk = OrderedDict()
            blk[f"{n}_convbatch"] = Conv2dBatchReLU(cout, cout // 2, 1, 1, 0); n += 1
            blk[f"{n}_convbatch"] = Conv2dBatchReLU(cout // 2, cout, 3, 1, 1); n += 1
            layers[f"{chr(letter)}_residual"] = Residual(blk); letter += 1
        cin = cout
    return layers


def _head(first, cin, mid, out_ch, conv_name):
    """Five alternating 1x1 / 3x3 units, one more 3x3 and the linear 1x1 predictor, numbered from `first`."""
    d = OrderedDict()
    n = first
    c = cin
    for i in range(6):
        co, k = (mid, 1) if i % 2 == 0 else (mid * 2, 3)
        d[f"{n}_convbatch"] = Conv2dBatchReLU(c, co, k, 1, k // 2); n += 1
        c = co
    d[conv_name] = HeadConv2d(c, out_ch, 1, 1, 0)              # NCHW head + structured backward (utils/headconv.py)
    return d


class YOLOv3(_Base):
    stride = (32, 16, 8)
    sample_anchors = [[(116, 90), (156, 198), (373, 326)], [(30, 61), (62, 45), (59, 119)], [(10, 13), (16, 30), (33, 23)]]
    img_size = 416
    grid_size = 0
    ignore_thres = 0.5
    anch_masks = None

    def __init__(self, classes, args):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "YOLOv3"
        self.anchors = [[(a[0] / s, a[1] / s) for a in self.sample_anchors[i]] for i, s in enumerate(self.stride)]   # :54-56
        no = 3 * (5 + self.num_classes)
        self.extractor = SelectiveSequential(["k_residual", "s_residual"], _darknet53())
        self.detector = nn.ModuleList([
            SelectiveSequential(["57_convbatch"], _head(53, 1024, 512, no, "59_conv")),
            nn.Sequential(OrderedDict([("60_convbatch", Conv2dBatchReLU(512, 256, 1, 1, 0)), ("61_upsample", _Upsample2x(scale_factor=2, mode="nearest"))])),
            SelectiveSequential(["66_convbatch"], _head(62, 256 + 512, 256, no, "68_conv")),
            nn.Sequential(OrderedDict([("69_convbatch", Conv2dBatchReLU(256, 128, 1, 1, 0)), ("70_upsample", _Upsample2x(scale_factor=2, mode="nearest"))])),
            nn.Sequential(_head(71, 128 + 256, 128, no, "77_conv")),
        ])
        self.sample = (1, 3, self.img_size, self.img_size)
        self.criterion = configure_loss(args, None, self.anchors, None, self.num_classes, self.img_size)
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)

    def forward(self, x):
        x, inter = self.extractor(x)
        o0, x = self.detector[0](x)
        o1, x = self.detector[2](torch.cat((self.detector[1](x), inter["s_residual"]), 1))
        o2 = self.detector[4](torch.cat((self.detector[3](x), inter["k_residual"]), 1))
        return [o0, o1, o2]

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
