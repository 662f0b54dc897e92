"""SSD300 (VGG-16 trunk + 5 extra stages + 6 class/box head pairs) as the CALLER of the hot path; mirrors
`model/SSD.py:16-310`: class name, ctor `(classes, args)`, `forward(x) -> (out_bbx [B,8732,4], out_cl [B,8732,C])`,
`.iou_boxes` (= `get_dboxes()`), `.criterion`, `.non_max_suppression`, and the reference's `state_dict` keys
(`f1.<i>`, `base1.<i>`, `f2..f6.<i>`, `cl1..cl6.0`, `bbx1..bbx6.0`).

The reference takes its trunk from `torchvision.models.vgg16(pretrained=True).features` (`SSD.py:46`); neither
torchvision nor its weight download exists here, so the trunk is laid out from the published VGG-16 "D" table with
the same Sequential indices (conv at 0,2,5,7,10,12,14,17,19,21 | 24,26,28; pools at 4,9,16,23,30) and random init.
The key/shape manifest in tests/golden/state_dict_ssd.json is therefore DERIVED from the reference's source text,
not captured from a run of it ("parity unpinned" for that manifest; loss, priors and NMS are pinned by ssd.npz / nms.npz).
Convolutions are plain `nn.Conv2d` (MIOpen); priors, matching + loss and NMS are the HIP path."""
import os

import torch
import torch.nn as nn

from ..LightningFunc import lightningUtils as _lu
from ..LightningFunc import optimizer as _opt
from ..LightningFunc import step as _step
from ..LightningFunc.losses import configure_loss
from ..LightningFunc.nms import greedy_non_max_suppression
from ..LightningFunc.utils import ssdhead as _ssdhead
from ..LightningFunc.utils.SSDUtils import get_dboxes

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module

# VGG-16 configuration D: channel widths, "M" = 2x2 max-pool
_VGG_D = [64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"]


def _vgg_features():
    layers, cin = [], 3
    for v in _VGG_D:
        if v == "M":
            layers.append(nn.MaxPool2d(2, 2))
        else:
            layers += [nn.Conv2d(cin, v, 3, padding=1), nn.ReLU(inplace=True)]
            cin = v
    layers[16] = nn.MaxPool2d(2, ceil_mode=True)         # 75 -> 38 (SSD.py:49)
    layers[-1] = nn.MaxPool2d(3, 1, padding=1)           # pool5: 3x3 stride 1 (SSD.py:50)
    return layers


def _extra(cin, mid, cout, **second):
    return nn.Sequential(nn.Conv2d(cin, mid, 1), nn.ReLU(inplace=True), nn.Conv2d(mid, cout, 3, **second), nn.ReLU(inplace=True))


class SSD(_Base):
    img_size = 300
    # (source channels, priors per cell) of the six prediction maps: 38, 19, 10, 5, 3, 1
    _sources = [(512, 4), (1024, 6), (512, 6), (256, 6), (256, 4), (256, 4)]

    def __init__(self, classes, args, device=None):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "SSD"
        self.size = (300, 300)
        vgg = _vgg_features()
        # attribute order = the reference's registration order (state_dict key order)
        self.f1 = nn.Sequential(*vgg[:23])
        self._heads(1)
        self.base1 = nn.Sequential(*vgg[23:])
        self.f2 = nn.Sequential(nn.Conv2d(512, 1024, 3, dilation=3, padding=3), nn.ReLU(inplace=True),
                                nn.Conv2d(1024, 1024, 1), nn.ReLU(inplace=True))
        self._heads(2)
        self.f3 = _extra(1024, 256, 512, stride=2, padding=1)
        self._heads(3)
        self.f4 = _extra(512, 128, 256, stride=2, padding=1)
        self._heads(4)
        self.f5 = _extra(256, 128, 256)
        self._heads(5)
        self.f6 = _extra(256, 128, 256)
        self._heads(6)
        for name in [f"{p}{i}" for i in range(1, 7) for p in ("cl", "bbx")] + [f"f{i}" for i in range(2, 7)]:
            for layer in getattr(self, name):                       # SSD.py:225-233 (the trunk keeps its own init)
                if isinstance(layer, nn.Conv2d):
                    nn.init.xavier_normal_(layer.weight)
                    nn.init.constant_(layer.bias, 0)
        self.sample = (1, 3, self.img_size, self.img_size)
        self.iou_boxes = None
        self.criterion = None
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)
        if device is not None:
            self._build_priors(torch.device(device))

    def _heads(self, i):
        ch, n = self._sources[i - 1]
        setattr(self, f"cl{i}", nn.Sequential(nn.Conv2d(ch, n * self.num_classes, 3, padding=1)))
        setattr(self, f"bbx{i}", nn.Sequential(nn.Conv2d(ch, n * 4, 3, padding=1)))

    def _build_priors(self, dev):
        """SSD.py:30-32: the 8732-prior table, built on the device, and the criterion bound to it."""
        self.iou_boxes = get_dboxes(device=dev)
        self.criterion = configure_loss(self.args, self.iou_boxes, None, None, self.num_classes, self.img_size)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        p = next(self.parameters())
        if p.is_cuda and (self.iou_boxes is None or self.iou_boxes.device != p.device):
            self._build_priors(p.device)
        return self

    @property
    def sampleImg(self):
        return torch.rand(self.sample, device=next(self.parameters()).device)

    def forward(self, x):
        if x.is_cuda and (self.iou_boxes is None or self.iou_boxes.device != x.device):
            self._build_priors(x.device)
        B = x.size(0)
        out_cl, out_bbx = [], []
        src = self.f1(x)
        cls_convs, box_convs = [getattr(self, f"cl{i}")[0] for i in range(1, 7)], [getattr(self, f"bbx{i}")[0] for i in range(1, 7)]
        if (_ssdhead.enabled and x.is_cuda and torch.is_grad_enabled() and not torch.is_autocast_enabled() and x.dtype == torch.float32 and all(len(getattr(self, f"{p}{i}")) == 1 for i in range(1, 7) for p in ("cl", "bbx"))
                and all(_ssdhead.is_hostable(c) for c in cls_convs + box_convs)):
            # the twelve head convolutions hosted in one autograd node: SSDLoss may then run their backward from the rows that carry a gradient
            sources = [src]
            for i in range(2, 7):
                src = self.f2(self.base1(src)) if i == 2 else getattr(self, f"f{i}")(src)
                sources.append(src)
            return _ssdhead.head_pairs(cls_convs, box_convs, sources, self.num_classes)
        for i in range(1, 7):
            if i == 2:
                src = self.f2(self.base1(src))
            elif i > 2:
                src = getattr(self, f"f{i}")(src)
            out_cl.append(getattr(self, f"cl{i}")(src).permute(0, 2, 3, 1).contiguous().view(B, -1, self.num_classes))
            out_bbx.append(getattr(self, f"bbx{i}")(src).permute(0, 2, 3, 1).contiguous().view(B, -1, 4))
        return torch.cat(out_bbx, 1), torch.cat(out_cl, 1)

    def non_max_suppression(self, predictions, topk=100, nms_thresh=0.5, class_thresh=0.45, mode="union"):
        return greedy_non_max_suppression(predictions, self.iou_boxes, topk, nms_thresh, class_thresh, mode)

    training_step = _step.training_step
    training_epoch_end = _step.training_epoch_end
    validation_step = _step.validation_step
    validation_epoch_end = _step.validation_epoch_end
    test_step = _step.test_step
    test_epoch_end = _step.test_epoch_end
    write_Best_model_path = _lu.write_Best_model_path
    read_Best_model_path = _lu.read_Best_model_path
    configure_optimizers = _opt.configure_optimizers
