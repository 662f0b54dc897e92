#!/usr/bin/env python3
"""Where does a one-launch BatchNorm kernel spend its time?  In-kernel readings of the 100 MHz clock at its phase boundaries
(odpl_bn_fused_debug_stamps), per workgroup; printed as medians / maxima relative to the first workgroup's start."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.nn as nn

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act

dev = torch.device("cuda:0")
L = _lib.lib()
_lib.tune_set("bn_fused", 1)          # opt-in path (off by default)
names = ["start", "sums", "drained", "peers_seen", "acquired", "stats", "stores_issued"]
for shp in [(16, 512, 13, 13), (16, 256, 26, 26), (16, 512, 26, 26), (16, 128, 52, 52)]:
    x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
    bn = nn.BatchNorm2d(shp[1]).to(dev).train()
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    for _ in range(3):
        bn_act(x, bn, "mish")
    big.sum()
    L.odpl_bn_fused_debug_stamps(1, None, 0)
    y = bn_act(x, bn, "mish")
    torch.cuda.synchronize()
    n = 2048
    buf = (ctypes.c_uint64 * (n * 8))()
    L.odpl_bn_fused_debug_stamps(0, buf, n)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 8).astype(np.int64)
    a = a[a[:, 0] != 0]
    a = a[a[:, 0] >= a[:, 0].max() - 10_000_000]          # the last launch only (older launches left older readings)
    t0 = a[:, 0].min()
    rel = (a[:, :7] - t0) / 100.0
    print(f"shape {shp} workgroups {a.shape[0]}")
    for i, nme in enumerate(names):
        c = rel[:, i]
        print(f"  {nme:14s} min {c.min():7.2f}  median {np.median(c):7.2f}  max {c.max():7.2f} us")
    d = np.diff(rel, axis=1)
    print("  phase medians:", " ".join(f"{names[i]}->{names[i + 1]}={np.median(d[:, i]):.2f}" for i in range(6)))
