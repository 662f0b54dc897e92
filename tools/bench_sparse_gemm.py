#!/usr/bin/env python3
"""The gathered GEMMs of csrc/odpl_sparsefwd.hip / odpl_sparseconv.hip against the library on a config-4 head unit
(Conv2d(256, 256, 3, 1, 1) on P3 of a 4 x 800 x 1333 batch: 66 800 pixels): forward and backward time by the share of listed pixels --
where the break-even density is, and what the kernels reach on a full list."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B, C, H, W = 4, 256, 100, 167
P = B * H * W
conv = nn.Conv2d(C, C, 3, 1, 1).to(dev).to(memory_format=torch.channels_last)
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, C, generator=g).to(dev).permute(0, 3, 1, 2).requires_grad_()
sc.max_density = 2.0


def timeit(fn, n=10):
    """wall clock around n calls, best of three rounds (a round now and then eats a one-off 40-60 ms allocator stall on fresh sizes)"""
    import time
    for _ in range(3):
        fn()
    best = None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n * 1e6
        best = dt if best is None or dt < best else best
    return best


lib_f = timeit(lambda: F.conv2d(x, conv.weight, conv.bias, 1, 1))
gy_full = torch.randn(B, H, W, C, generator=g).to(dev).permute(0, 3, 1, 2)
lib_b = timeit(lambda: torch.autograd.grad(F.conv2d(x, conv.weight, conv.bias, 1, 1), [x, conv.weight, conv.bias], gy_full)) - lib_f
print(json.dumps({"route": "library (solver search)", "forward_us": round(lib_f, 1), "backward_us": round(lib_b, 1),
                  "forward_TFLOPs": round(2 * P * 9 * C * C / lib_f / 1e6, 1)}))
for share in (0.002, 0.01, 0.03, 0.1, 0.3, 1.0):
    n = max(1, int(share * P))
    sel = torch.randperm(P, generator=g)[:n].sort().values.to(torch.int32)
    lst = torch.zeros(P, dtype=torch.int32, device=dev)
    lst[:n] = sel.to(dev)
    cnt = torch.tensor([n], dtype=torch.int32, device=dev)
    fwd = timeit(lambda: sc.forward_on_pixels(x, conv.weight, conv.bias, sc.ACT["relu"], (lst, cnt)))
    on = torch.zeros(P, dtype=torch.bool, device=dev)
    on[sel.long().to(dev)] = True
    gy = gy_full * on.view(B, 1, H, W)
    y = sc.conv_bias_act(x, conv, "relu", (lst, cnt))
    bwd = timeit(lambda: torch.autograd.grad(y, [x, conv.weight, conv.bias], gy, retain_graph=True))
    print(json.dumps({"route": "listed pixels", "share": share, "pixels": n, "forward_us": round(fwd, 1), "backward_us": round(bwd, 1),
                      "forward_TFLOPs": round(2 * n * 9 * C * C / fwd / 1e6, 1)}))
