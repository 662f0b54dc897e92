#!/usr/bin/env python3
"""The eval-side NMS launches at BASELINE sizes, to be run under `rocprofv3 --kernel-trace --stats`: config 2 (YOLO merge NMS, 16 x 10 647
candidates, 80 classes) and config 4 (greedy top-100 NMS, 4 x 200 700 anchors)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.LightningFunc import nms
from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
B = 16
hn = []
for G in (52, 26, 13):
    h = torch.randn(B, 3, 85, G, G, generator=g)
    h[:, :, 0:2] = torch.rand(B, 3, 2, 1, 1, generator=g) * 300 + 50 + torch.randn(B, 3, 2, G, G, generator=g) * 40
    h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 80 + 20
    h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.8 - 1.0
    h[:, :, 5:] = torch.rand(B, 3, 80, G, G, generator=g)
    hn.append(h.view(B, 255, G, G).contiguous().to(dev))
anchors = get_anchor_boxes([1333, 800], dev)
An = anchors.shape[0]
loc = (torch.randn(4, An, 4, generator=g) * 0.3).to(dev)
cls = (torch.randn(4, An, 80, generator=g) - 3.0).to(dev)
for _ in range(int(os.environ.get("N", 20))):
    nms.yolo_non_max_suppression(hn)
    if os.environ.get("ONE_WG"):
        nms.yolo_non_max_suppression(hn, one_workgroup=True)
    nms.greedy_non_max_suppression((loc, cls), anchors)
torch.cuda.synchronize()
