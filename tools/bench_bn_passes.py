#!/usr/bin/env python3
"""The four BN+activation launches (stats-reduce, apply, backward-reduce, dx) on ONE YOLOv4 stage shape, for rocprofv3:
SHAPE=0..5 picks the shape, TB / SB set the launch-geometry knobs.  Cold caches (1 GiB read between calls) unless COLD=0."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act

dev = torch.device("cuda:0")
shapes = [(16, 32, 416, 416), (16, 64, 208, 208), (16, 128, 104, 104), (16, 256, 52, 52), (16, 512, 26, 26), (16, 1024, 13, 13)]
_s = os.environ.get("SHAPE", "0")
shp = tuple(int(v) for v in _s.split(",")) if "," in _s else shapes[int(_s)]
if "TB" in os.environ:
    _lib.tune_set("bn_target_blocks", int(os.environ["TB"]))
if "SB" in os.environ:
    _lib.tune_set("bn_stream_max_blocks", int(os.environ["SB"]))
if "RT" in os.environ:
    _lib.tune_set("bn_max_rows_per_tile", int(os.environ["RT"]))
cold = os.environ.get("COLD", "1") == "1"
big = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None
x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
w = torch.randn_like(x)
bn = nn.BatchNorm2d(shp[1]).to(dev).train()
for it in range(12):
    xi = x.clone().requires_grad_()
    if cold:
        big.sum()
    y = bn_act(xi, bn, "mish")
    if cold:
        big.sum()
    y.backward(w)
torch.cuda.synchronize()
print("done", shp)
