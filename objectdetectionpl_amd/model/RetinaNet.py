"""RetinaNet (ResNet-50 FPN + two 4-conv heads) as the CALLER of the hot path; mirrors `model/RetinaNet.py:44-178`
and `model/backbone/RetinaNetbone.py`: class name, ctor `(classes, args)`, `forward(x) -> (loc_preds [B,An,4],
cls_preds [B,An,C])`, `.iou_boxes`, `.criterion`, `.non_max_suppression`, identical `state_dict` keys.
Convolutions are plain `nn.Conv2d` (MIOpen); BN+ReLU epilogues use the fused HIP kernels on channels-last inputs."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from ..LightningFunc import lightningUtils as _lu
from ..LightningFunc import optimizer as _opt
from ..LightningFunc import step as _step
from ..LightningFunc.losses import configure_loss
from ..LightningFunc.nms import greedy_non_max_suppression
from ..LightningFunc.utils import bnact as _bnact
from ..LightningFunc.utils import retinahead as _rh
from ..LightningFunc.utils import sparseconv as _sc
from ..LightningFunc.utils.RetinaUtils import get_anchor_boxes

try:
    import pytorch_lightning as pl
    _Base = pl.LightningModule
except Exception:  # noqa: BLE001
    _Base = nn.Module


def _bn(x, bn, act):
    if _bnact.fusable(x, bn):
        return _bnact.bn_act(x, bn, act)
    y = bn(x)
    return F.relu(y) if act == "relu" else y


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, in_planes, planes, stride=1):
        super().__init__()
        out = self.expansion * planes
        self.conv1 = nn.Conv2d(in_planes, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False); self.bn3 = nn.BatchNorm2d(out)
        self.downsample = nn.Sequential()
        if stride != 1 or in_planes != out:
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, out, 1, stride, bias=False), nn.BatchNorm2d(out))

    def forward(self, x):
        y = _bn(self.conv1(x), self.bn1, "relu")
        y = _bn(self.conv2(y), self.bn2, "relu")
        s = _bn(self.downsample[0](x), self.downsample[1], "linear") if len(self.downsample) else x
        z = self.conv3(y)
        if _bnact.fusable(z, self.bn3) and z.dtype == torch.float32:
            return _bnact.bn_add_act(z, self.bn3, s, "relu")          # relu(bn3(z) + s) in one write
        return F.relu(_bn(z, self.bn3, "linear") + s)


class FPN(nn.Module):
    def __init__(self, block, num_blocks):
        super().__init__()
        self.in_planes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        for i, (planes, stride) in enumerate(((64, 1), (128, 2), (256, 2), (512, 2)), 1):
            setattr(self, f"layer{i}", self._stage(block, planes, num_blocks[i - 1], stride))
        self.conv6 = nn.Conv2d(2048, 256, 3, 2, 1)
        self.conv7 = nn.Conv2d(256, 256, 3, 2, 1)
        self.latlayer1 = nn.Conv2d(2048, 256, 1)
        self.latlayer2 = nn.Conv2d(1024, 256, 1)
        self.latlayer3 = nn.Conv2d(512, 256, 1)
        self.toplayer1 = nn.Conv2d(256, 256, 3, 1, 1)
        self.toplayer2 = nn.Conv2d(256, 256, 3, 1, 1)

    def _stage(self, block, planes, n, stride):
        layers = []
        for s in [stride] + [1] * (n - 1):
            layers.append(block(self.in_planes, planes, s))
            self.in_planes = planes * block.expansion
        return nn.Sequential(*layers)

    @staticmethod
    def _up_add(x, y):
        return F.interpolate(x, size=y.shape[2:], mode="bilinear", align_corners=False) + y     # F.upsample(bilinear), :57

    def forward(self, x, support=None, reach=0):
        """support (utils/retinahead.PixelSupport) + reach: a loss-driven training step -- the heads will read p3 / p4 within `reach`
        pixels of a positive anchor only (on the levels the support lists), so the top-down path is evaluated there only: p_l on
        N_l = reach(l) united with what the finer level's merge interpolates from, the merged map one pixel further out (the
        smoothing convolution's reads), the lateral 1 x 1 convolution where the merge is."""
        return self.top_down(*self.bottom_up(x), support=support, reach=reach)

    def bottom_up(self, x):
        """The ResNet stages (every BatchNorm of the network lives here): c3, c4, c5."""
        c1 = F.max_pool2d(_bn(self.conv1(x), self.bn1, "relu"), 3, 2, 1)
        c2 = self.layer1(c1); c3 = self.layer2(c2); c4 = self.layer3(c3); c5 = self.layer4(c4)
        return c3, c4, c5

    def top_down(self, c3, c4, c5, support=None, reach=0):
        """p3 .. p7 from the stage outputs (RetinaNetbone.py:55-75); no BatchNorm, so evaluating it twice on the same stage outputs
        changes no buffer."""
        p6 = self.conv6(c5)
        p7 = self.conv7(F.relu(p6))
        p5 = self.latlayer1(c5)
        if support is not None and support.has(0) and self._topdown_on_pixels_ok(c3, c4, p5):
            return self._topdown_on_pixels(c3, c4, p5, support, reach) + (p5, p6, p7)
        p4 = self._smooth(self.toplayer1, self._up_add(p5, self.latlayer2(c4)))
        p3 = self._smooth(self.toplayer2, self._up_add(p4, self.latlayer3(c3)))
        return p3, p4, p5, p6, p7

    def _topdown_on_pixels_ok(self, c3, c4, p5):
        return (_sc._applicable(c3, self.latlayer3) and _sc.conv_ok(self.toplayer2) and self.toplayer2.in_channels == self.latlayer3.out_channels
                and p5.dtype == torch.float32 and p5.is_contiguous(memory_format=torch.channels_last) and p5.shape[1] == self.latlayer3.out_channels)

    def _topdown_on_pixels(self, c3, c4, p5, sup, reach):
        B = c3.shape[0]
        (H3, W3), (H4, W4) = c3.shape[2:], c4.shape[2:]
        # P3: p3 where the heads read it, the merged map one pixel further out
        need_p3 = sup.pair(0, reach)
        need_m3 = sup.pair(0, reach + 1)
        # P4: what P3's merge interpolates from, plus (if P4's heads run on lists too) what they read; else every pixel (dense P4)
        if (sup.has(1) and _sc._applicable(c4, self.latlayer2) and _sc.conv_ok(self.toplayer1)
                and self.toplayer1.in_channels == self.latlayer2.out_channels == p5.shape[1]):
            need_p4, need_m4 = _sc.union_lists(B, H4, W4, own=sup.pair(1, reach), finer=need_m3, finer_hw=(H3, W3))
            l4 = _sc.conv_bias_act(c4, self.latlayer2, "linear", need_m4)
            m4 = _sc.up_add_on_pixels(p5, l4, need_m4)
            p4 = _sc.conv_bias_act(m4, self.toplayer1, "linear", need_p4)
        else:
            p4 = self._smooth(self.toplayer1, self._up_add(p5, self.latlayer2(c4)))
        l3 = _sc.conv_bias_act(c3, self.latlayer3, "linear", need_m3)
        m3 = _sc.up_add_on_pixels(p4, l3, need_m3)
        p3 = _sc.conv_bias_act(m3, self.toplayer2, "linear", need_p3)
        return p3, p4

    @staticmethod
    def _smooth(conv, x):
        """toplayer1 / toplayer2 (RetinaNetbone.py: Conv2d(256, 256, 3, 1, 1) on the merged maps): their output gradient is the sum of the two
        head towers' input gradients -- still a few per cent of the pixels on the large levels -- so their backward also walks the
        active pixels only (utils/sparseconv.py; act = linear)"""
        from ..LightningFunc.utils import sparseconv as _sc
        return _sc.conv_bias_act(x, conv, "linear") if _sc.applicable(x, conv) else conv(x)


class RetinaNet(_Base):
    num_anchors = 9
    img_size = 600

    def __init__(self, classes, args, device=None):
        super().__init__()
        self.classes = classes
        self.num_classes = len(classes)
        self.args = args
        self.backbone = "RetinaNet"
        self.fpn = FPN(Bottleneck, [3, 4, 6, 3])
        self.loc_head = self._make_head(self.num_anchors * 4)
        self.cls_head = self._make_head(self.num_anchors * self.num_classes)
        self.sample = (1, 3, self.img_size, self.img_size)
        self._anchor_device = device
        self.iou_boxes = None
        self.criterion = None
        self.checkname = self.backbone
        self.dir = os.path.join("log_dir", getattr(args, "data_module", "synthetic"), self.checkname)
        if device is not None:
            self._build_anchors(torch.device(device))

    def _build_anchors(self, dev):
        """RetinaNet.py:57-59: the anchor table for (img_size, img_size), built on the device."""
        self.iou_boxes = get_anchor_boxes(torch.Tensor([self.img_size, self.img_size]), dev)
        self.criterion = configure_loss(self.args, self.iou_boxes, None, None, self.num_classes, self.img_size)

    def _apply(self, fn, *a, **k):
        super()._apply(fn, *a, **k)
        p = next(self.parameters())
        if p.is_cuda and (self.iou_boxes is None or self.iou_boxes.device != p.device):
            self._build_anchors(p.device)        # the reference builds them in the ctor with .cuda() (RetinaNet.py:57-59)
        return self

    @staticmethod
    def _make_head(out_planes):
        layers = []
        for _ in range(4):
            layers += [nn.Conv2d(256, 256, 3, 1, 1), nn.ReLU(True)]
        layers.append(nn.Conv2d(256, out_planes, 3, 1, 1))
        return nn.Sequential(*layers)

    @staticmethod
    def _run_tower(head, fm, support=None, level=0):
        """the four [Conv2d(256,256,3,1,1), ReLU] units of a head (RetinaNet.py:93-99): MIOpen convolution + one fused epilogue each
        (utils/bnact.py:conv_bias_act).  support (utils/retinahead.PixelSupport, a loss-driven training step): on the levels it lists,
        unit t of n is evaluated within n - t pixels of a positive anchor only -- what the units after it and the criterion will read."""
        mods = list(head)
        n = (len(mods) - 1) // 2
        for t, i in enumerate(range(0, len(mods) - 1, 2)):
            # on the large levels the backward of conv + bias + ReLU walks the pixels that carry a gradient only (utils/sparseconv.py)
            if _sc.applicable(fm, mods[i]):
                sup = support.pair(level, n - t) if support is not None and support.has(level) else None
                fm = _sc.conv_bias_act(fm, mods[i], "relu", sup)
            else:
                fm = _bnact.conv_bias_act(fm, mods[i], "relu")
        return fm

    @classmethod
    def _run_head(cls, head, fm):
        """[Conv2d(256,256,3,1,1), ReLU] x 4 + Conv2d (RetinaNet.py:93-99)"""
        if not fm.is_cuda:
            return head(fm)
        return list(head)[-1](cls._run_tower(head, fm))

    def forward(self, x):
        if x.is_cuda and (self.iou_boxes is None or self.iou_boxes.device != x.device):
            self._build_anchors(x.device)          # the anchor table (criterion / NMS side) lives on the HIP device only
        return self._heads(x, self.fpn(x))

    def _heads(self, x, feats):
        if (x.is_cuda and _rh.enabled and self.num_anchors == 9 and _rh.is_hostable(self.loc_head[-1]) and _rh.is_hostable(self.cls_head[-1])
                and feats[0].dtype == torch.float32):
            # the two final convolutions of all levels hosted together: RetinaNetLoss can then run their backward from its structured
            # gradient (utils/retinahead.py); same values as the loop below
            xl = [self._run_tower(self.loc_head, fm) for fm in feats]
            xc = [self._run_tower(self.cls_head, fm) for fm in feats]
            return _rh.final_convs(self.loc_head[-1], self.cls_head[-1], xl, xc)
        loc, cls = [], []
        for fm in feats:
            loc.append(self._run_head(self.loc_head, fm).permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, 4))
            cls.append(self._run_head(self.cls_head, fm).permute(0, 2, 3, 1).contiguous().view(x.size(0), -1, self.num_classes))
        return torch.cat(loc, 1), torch.cat(cls, 1)

    topdown_on_pixels = True       # class switch: the FPN's top-down path under the same restriction (A/B: --dense-topdown)
    loss_driven_step = True        # class switch (tools/bench_retina_step.py --dense-forward): False = training_step is forward + criterion

    def loss_driven_mask(self, x, y):
        """Levels of the heads a training step may evaluate on the criterion's pixels only; 0 = forward + criterion.  Decided from
        shapes alone, before anything runs: training mode with gradients on, fp32 outside autocast, the criterion is exactly the fused
        RetinaNetLoss over THIS input's anchor table, both heads hostable with four units; a level qualifies from
        `sparseconv.min_pixels` pixels on."""
        from ..LightningFunc.losses_dense import RetinaNetLoss
        crit = self.criterion
        if not (self.loss_driven_step and self.training and torch.is_grad_enabled() and torch.is_tensor(x) and x.is_cuda and x.dim() == 4
                and x.dtype == torch.float32 and not torch.is_autocast_enabled() and torch.is_tensor(y) and y.is_cuda
                and _rh.enabled and _sc.enabled and type(crit) is RetinaNetLoss and crit.fuse_head_backward
                and self.num_anchors == 9 and crit.num_classes == self.num_classes and len(self.loc_head) == 9 and len(self.cls_head) == 9
                and _rh.is_hostable(self.loc_head[-1]) and _rh.is_hostable(self.cls_head[-1])
                and self.loc_head[0].weight.dtype == torch.float32):
            return 0
        shapes = self._level_shapes(x)
        if 9 * sum(h * w for h, w in shapes) != len(crit.iou_boxes):
            return 0
        mask = 0
        for l, (h, w) in enumerate(shapes):
            if x.shape[0] * h * w >= _sc.min_pixels and h > 1 and w > 1:
                mask |= 1 << l
        return mask & ~self._crowded_levels()

    dense_share = 0.5              # a level whose heads would be evaluated on more than this share of its pixels goes back to the library
    reprobe_every = 200            # ... and is tried again after this many steps

    def _crowded_levels(self):
        """Levels the last looks found too crowded for lists to pay (tools/bench_sparse_gemm.py: the gathered GEMM breaks even with the
        library at about half of the pixels).  The pixel counts are read back WITHOUT a sync: `forward_and_loss` starts a pinned copy
        of PixelSupport's counters, this looks at it on a later step."""
        st = self.__dict__.setdefault("_support_stats", {"pending": None, "crowded": 0, "since": 0})
        if st["pending"] is not None:
            pinned, ev, pixels, reach, depth = st["pending"]
            if ev.query():
                st["pending"] = None
                crowded = 0
                for l, n in enumerate(pixels):
                    if n and int(pinned[l * depth + reach]) > self.dense_share * n:
                        crowded |= 1 << l
                if crowded != st["crowded"]:
                    st["crowded"], st["since"] = crowded, 0
        if st["crowded"]:
            st["since"] += 1
            if st["since"] >= self.reprobe_every:
                st["crowded"], st["since"] = 0, 0
        return st["crowded"]

    def _watch_support(self, sup, reach):
        st = self.__dict__.setdefault("_support_stats", {"pending": None, "crowded": 0, "since": 0})
        if st["pending"] is None and not torch.cuda.is_current_stream_capturing():
            pinned = torch.empty(sup.counts.numel(), dtype=torch.int32, pin_memory=True)
            pinned.copy_(sup.counts, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            st["pending"] = (pinned, ev, [p if sup.has(l) else 0 for l, p in enumerate(sup.pix)], reach, sup.depth)

    @staticmethod
    def _level_shapes(x):
        """P3 .. P7 of an [B, 3, H, W] input: every stride-2 stage of the FPN (7x7 s2 p3, 3x3 pool s2 p1, 3x3 s2 p1) halves rounding up"""
        H, W = x.shape[2], x.shape[3]
        return [(-(-H // 2 ** (3 + l)), -(-W // 2 ** (3 + l))) for l in range(5)]

    def forward_and_loss(self, x, y, mask=None):
        """`criterion(forward(x), y)` (step.py:17-18) for a training step, or None when the step has to be forward + criterion
        (`loss_driven_mask`; nothing has run then; once the route is taken the backbone runs exactly once, whatever the FPN then
        turns out to deliver).  RetinaNetLoss reads the positive anchor rows of the predictions and nothing else
        (losses.py:336-339, 458-461), and which rows those are depends on anchors and boxes only: the assignment runs first and the
        heads are evaluated where the loss will look (csrc/odpl_sparsefwd.hip) -- same loss, same gradients, the predictions of the
        other rows are never formed."""
        if mask is None:                             # (a data-parallel caller decides once per step and passes its decision: trainer._ForwardVia)
            mask = self.loss_driven_mask(x, y)
        if not mask:
            return None
        if self.iou_boxes is None or self.iou_boxes.device != x.device:
            self._build_anchors(x.device)
        crit = self.criterion
        n_units = (len(self.loc_head) - 1) // 2
        pos, npos, token = crit.assign(y, x.shape[0])                       # needs anchors and boxes only
        sup = _rh.PixelSupport(pos, npos, x.shape[0], self._level_shapes(x), n_units + 3, mask)      # reach 0 .. n_units + 2
        self._watch_support(sup, n_units)
        stages = self.fpn.bottom_up(x)
        feats = self.fpn.top_down(*stages, support=sup, reach=n_units + 1) if self.topdown_on_pixels else self.fpn.top_down(*stages)
        ok = [(f.shape[2], f.shape[3]) for f in feats] == self._level_shapes(x) and feats[0].dtype == torch.float32
        for l, f in enumerate(feats):
            if (mask >> l) & 1 and not (_sc._applicable(f, self.loc_head[0]) and _sc._applicable(f, self.cls_head[0])):
                ok = False
        if not ok:                                   # (an FPN that does not produce what the shapes promised): the dense heads on the SAME
            crit._assignment = None                  # stage outputs -- the ResNet (and with it every BatchNorm buffer update) runs once per step
            return crit(self._heads(x, self.fpn.top_down(*stages)), y)
        xl = [self._run_tower(self.loc_head, fm, sup, l) for l, fm in enumerate(feats)]
        xc = [self._run_tower(self.cls_head, fm, sup, l) for l, fm in enumerate(feats)]
        out = _rh.final_convs(self.loc_head[-1], self.cls_head[-1], xl, xc, support=sup, assignment=token)
        return crit(out, y)

    def freeze_bn(self):
        for m in self.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.eval()

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
