#!/usr/bin/env python3
"""yolo_decode_kernel and yolo_nms_prep_kernel at config 2 (B=16, C=80, 416^2), cold caches, library event timing."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc import nms
from objectdetectionpl_amd.LightningFunc.utils import YoloV4Utils

dev = torch.device("cuda:0")
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
g = torch.Generator().manual_seed(3)
B, C = 16, 80
A18 = [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401]
M3 = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
heads = [torch.randn(B, 255, G, G, generator=g).to(dev) for G in (52, 26, 13)]
cells = sum(B * 3 * G * G for G in (52, 26, 13))
for name, fn, nbytes in (("yolo_decode", lambda: YoloV4Utils.decode_heads(heads, C, A18, M3, [8, 16, 32]), 4 * cells * 85 + 4 * cells * 84),
                         ("yolo_nms_prep", lambda: nms.yolo_non_max_suppression(heads), 4 * cells * 85 + cells * 40)):
    for _ in range(3):
        fn()
    _lib.profile_enable(name, True)
    for _ in range(15):
        big.sum()
        fn()
    torch.cuda.synchronize()
    _lib.profile_enable(name, False)
    ms = sorted(_lib.profile_collect(name)[0])
    us = 1e3 * ms[len(ms) // 2]
    print(json.dumps({"kernel": name, "median_us": round(us, 2), "frac_of_8TBps": round(nbytes / (us * 1e-6) / 8e12, 4)}))
