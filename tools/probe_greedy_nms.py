#!/usr/bin/env python3
"""Per-phase wall-clock stamps of greedy_nms_image_kernel at config 4's size (4 x 200 700 anchors, 80 classes, tools/prof_nms.py's
inputs); needs a library built with -DODPL_NMS_TRACE.  One JSON line: the phases' mean durations over the images, in us."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc import nms
from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(3)
anchors = get_anchor_boxes([1333, 800], dev)
An = anchors.shape[0]
loc = (torch.randn(4, An, 4, generator=g) * 0.3).to(dev)
cls = (torch.randn(4, An, 80, generator=g) - 3.0).to(dev)
L = _lib.lib()
L.odpl_greedy_nms_trace.argtypes = [ctypes.c_void_p]
buf = torch.zeros(4 * 16, dtype=torch.int64, device=dev)
for _ in range(5):
    out = nms.greedy_non_max_suppression((loc, cls), anchors)
torch.cuda.synchronize()
assert L.odpl_greedy_nms_trace(buf.data_ptr()) == 0
out = nms.greedy_non_max_suppression((loc, cls), anchors)
torch.cuda.synchronize()
L.odpl_greedy_nms_trace(None)
t = buf.cpu().numpy().reshape(4, 16).astype(np.float64) / 100.0
names = ["count", "key loads issued", "bound from the thread maxima", "list of keys at or above it", "order by counting + decode", "(general way)", "(general way: sort + decode)", "overlap matrix", "greedy loop", "output rows"]
d = {"kernel_span_us": round(float(t[:, 9].max() - t[:, 0].min()), 2)}
for i in range(1, 10):
    prev = np.where(t[:, i - 1] > 0, t[:, i - 1], np.nan)
    d[names[i] + "_us"] = round(float(np.nanmean(t[:, i] - prev)), 2)
print(json.dumps(d))
