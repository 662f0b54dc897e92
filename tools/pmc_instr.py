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
torch.cuda.synchronize()
print("done")
