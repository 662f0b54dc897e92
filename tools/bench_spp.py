#!/usr/bin/env python3
"""GPU time of the SPP block (fwd, bwd) at the YOLOv4 bench shape, channels-last and NCHW, vs the torch composition."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from objectdetectionpl_amd.LightningFunc.utils.spp import spp_cat


def ref(x):
    return torch.cat([F.max_pool2d(x, 13, 1, 6), F.max_pool2d(x, 9, 1, 4), F.max_pool2d(x, 5, 1, 2), x], 1)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("layout,impl,fwd_us,fwd_bwd_us")
for cl in (True, False):
    x = torch.randn(16, 512, 13, 13, device="cuda")
    if cl:
        x = x.contiguous(memory_format=torch.channels_last)
    w = torch.randn(16, 2048, 13, 13, device="cuda")
    if cl:
        w = w.contiguous(memory_format=torch.channels_last)
    for name, f in (("hip", spp_cat), ("torch", ref)):
        xr = x.clone().requires_grad_()
        with torch.no_grad():
            tf = timeit(lambda: f(xr))
        def fb():
            xr.grad = None
            f(xr).backward(w)
        print(f"{'channels_last' if cl else 'nchw'},{name},{tf:.1f},{timeit(fb):.1f}", flush=True)
