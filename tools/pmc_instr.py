#!/usr/bin/env python3
"""Drives the kernels DESIGN calls instruction-bound rather than HBM-bound (BN backward reduce with Mish, SPP forward / backward, Focus)
once each at their BASELINE shapes, for `rocprofv3 --pmc SQ_*` (tools/pmc_instr.sh): VERDICT r03 item 9."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act
from objectdetectionpl_amd.LightningFunc.utils.s2d import focus
from objectdetectionpl_amd.LightningFunc.utils.spp import spp_cat

dev = torch.device("cuda:0")
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
for shp in ((16, 32, 416, 416), (16, 512, 26, 26)):                  # the largest map and a launch-bound one
    x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn_like(x)
    bn = nn.BatchNorm2d(shp[1]).to(dev).train()
    for _ in range(4):
        xi = x.clone().requires_grad_()
        big.sum()
        y = bn_act(xi, bn, "mish")
        big.sum()
        y.backward(w)
    del x, w
x = torch.randn(16, 512, 13, 13, device=dev).contiguous(memory_format=torch.channels_last)
g = torch.randn(16, 2048, 13, 13, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(4):
    xi = x.clone().requires_grad_()
    big.sum()
    y = spp_cat(xi)
    big.sum()
    y.backward(g)
img = torch.rand(32, 3, 640, 640, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(4):
    big.sum()
    focus(img)
# the eval-side prep kernel, the bias + ReLU backward of RetinaNet's tower units, the dense YOLOv5 gradient stream (interleaved bf16 heads)
from objectdetectionpl_amd.LightningFunc import nms
from objectdetectionpl_amd.LightningFunc.losses_dense import MultiScaleRegionLoss_v5
from objectdetectionpl_amd.LightningFunc.utils.bnact import conv_bias_act
g = torch.Generator().manual_seed(3)
hn = [torch.randn(16, 255, G, G, generator=g).to(dev) for G in (52, 26, 13)]
for _ in range(4):
    big.sum()
    nms.yolo_non_max_suppression(hn)
conv = nn.Conv2d(256, 256, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
xt = torch.randn(4, 256, 100, 167, device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(4):
    xi = xt.clone().requires_grad_()
    y = conv_bias_act(xi, conv, "relu")
    big.sum()
    y.backward(torch.randn_like(y))
V5A = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]
B, C = 32, 80
n = B * 7
wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
t5 = torch.cat([torch.arange(B).repeat_interleave(7).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)
crit = MultiScaleRegionLoss_v5(V5A, None, None, None, None, C, 640)
prev = MultiScaleRegionLoss_v5.fuse_head_backward
MultiScaleRegionLoss_v5.fuse_head_backward = False           # the dense gradient stream (v5_expand) is what this pass counts
raws = [torch.randn(B, 3 * (5 + C), G, G, generator=g).to(dev).bfloat16().contiguous(memory_format=torch.channels_last).requires_grad_()
        for G in (80, 40, 20)]                              # the Detect convolutions' own layout: Yolo_Layers hands the loss views
for _ in range(4):
    for r in raws:
        r.grad = None
    big.sum()
    heads = [r.view(B, 3, 5 + C, r.shape[2], r.shape[3]).permute(0, 1, 3, 4, 2) for r in raws]
    crit(heads, t5)["loss"].backward()
MultiScaleRegionLoss_v5.fuse_head_backward = prev
torch.cuda.synchronize()
print("done")
