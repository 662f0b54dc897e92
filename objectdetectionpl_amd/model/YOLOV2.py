"""YOLOv2 (Darknet-19 + passthrough "reorg") as the CALLER of the hot path; mirrors `model/YOLOV2.py:17-222`:
class name, ctor `(classes, args)`, 5 grid-unit anchors, `forward(x) -> [B, 5*(5+C), G, G]`, `RegionLoss_v2`
criterion, identical `state_dict` keys (`stage1_conv<k>.0/1`, `stage2_a_conv<k>`, `stage2_b_conv`, `stage3_conv<k>`).
The network is a table; BN + LeakyReLU epilogues use the fused HIP kernels on channels-last fp32 activations."""
import os

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
from ..LightningFunc.utils.s2d import reorg as _reorg

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module

# (name, cin, cout, kernel, maxpool-after)
_STAGE1 = [("stage1_conv1", 3, 32, 3, True), ("stage1_conv2", 32, 64, 3, True), ("stage1_conv3", 64, 128, 3, False),
           ("stage1_conv4", 128, 64, 1, False), ("stage1_conv5", 64, 128, 3, True), ("stage1_conv6", 128, 256, 3, False),
           ("stage1_conv7", 256, 128, 1, False), ("stage1_conv8", 128, 256, 3, True), ("stage1_conv9", 256, 512, 3, False),
           ("stage1_conv10", 512, 256, 1, False), ("stage1_conv11", 256, 512, 3, False), ("stage1_conv12", 512, 256, 1, False),
           ("stage1_conv13", 256, 512, 3, False)]
_STAGE2A = [("stage2_a_conv1", 512, 1024, 3), ("stage2_a_conv2", 1024, 512, 1), ("stage2_a_conv3", 512, 1024, 3),
            ("stage2_a_conv4", 1024, 512, 1), ("stage2_a_conv5", 512, 1024, 3), ("stage2_a_conv6", 1024, 1024, 3),
            ("stage2_a_conv7", 1024, 1024, 3)]


def _unit(cin, cout, k, pool=False):
    layers = [nn.Conv2d(cin, cout, k, 1, k // 2, bias=False), nn.BatchNorm2d(cout), nn.LeakyReLU(0.1, inplace=True)]
    if pool:
        layers.append(nn.MaxPool2d(2, 2))
    return nn.Sequential(*layers)


def _run(seq, x):
    """conv -> fused BN+LeakyReLU (HIP) when the layout allows, else the plain modules; then the optional pool."""
    z = seq[0](x)
    if _bnact.fusable(z, seq[1]):
        y = _bnact.bn_act(z, seq[1], "leaky")
    else:
        y = seq[2](seq[1](z))
    return seq[3](y) if len(seq) > 3 else y


class YOLOv2(_Base):
    anchors = [(1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071)]
    img_size = 416
    grid_size = 0
    ignore_thres = 0.5
    anch_masks = None

    def __init__(self, classes, args):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "YOLOv2"
        for name, ci, co, k, pool in _STAGE1:
            setattr(self, name, _unit(ci, co, k, pool))
        self.stage2_a_maxpl = nn.MaxPool2d(2, 2)
        for name, ci, co, k in _STAGE2A:
            setattr(self, name, _unit(ci, co, k))
        self.stage2_b_conv = _unit(512, 64, 1)
        self.stage3_conv1 = _unit(256 + 1024, 1024, 3)
        self.stage3_conv2 = HeadConv2d(1024, len(self.anchors) * (5 + self.num_classes), 1, 1, 0, bias=False)
        self.sample = (1, 3, self.img_size, self.img_size)
        self.criterion = configure_loss(args, None, self.anchors, None, self.num_classes, self.img_size)
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)

    def forward(self, input):
        x = input
        for name, *_ in _STAGE1:
            x = _run(getattr(self, name), x)
        residual = x
        y = self.stage2_a_maxpl(x)
        for name, *_ in _STAGE2A:
            y = _run(getattr(self, name), y)
        z = _run(self.stage2_b_conv, residual)
        B, C, H, W = z.shape                                     # "reorg": YOLOV2.py:147-151
        if z.is_cuda and z.dtype == torch.float32 and C % 4 == 0 and H % 2 == 0 and W % 2 == 0:
            z = _reorg(z, y.is_contiguous(memory_format=torch.channels_last) and not y.is_contiguous())
        else:
            z = z.contiguous().view(B, C // 4, H, 2, W, 2).permute(0, 3, 5, 1, 2, 4).contiguous().view(B, -1, H // 2, W // 2)
        return self.stage3_conv2(_run(self.stage3_conv1, torch.cat((y, z), 1)))

    def non_max_suppression(self, predictions, conf_thres=0.5, nms_thres=0.4):
        return yolo_non_max_suppression(predictions, nms_thres=nms_thres, num_anchors=len(self.anchors))

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
