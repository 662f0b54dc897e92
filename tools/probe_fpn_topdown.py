#!/usr/bin/env python3
"""What the FPN's top-down path costs at BASELINE config 4 (B = 4, 800 x 1333), op by op, library kernels with the solver search:
the part of the loss-driven RetinaNet step that is still evaluated on every pixel of P3 / P4."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
B = 4


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def cl(*shape):
    return torch.randn(*shape, device=dev).contiguous(memory_format=torch.channels_last)


for name, cin, k, (h, w) in (("toplayer2 P3", 256, 3, (100, 167)), ("toplayer1 P4", 256, 3, (50, 84)), ("latlayer3 P3", 512, 1, (100, 167)),
                             ("latlayer2 P4", 1024, 1, (50, 84))):
    conv = nn.Conv2d(cin, 256, k, 1, k // 2).to(dev).to(memory_format=torch.channels_last)
    x = cl(B, cin, h, w).requires_grad_()
    y = conv(x)
    g = torch.randn_like(y)
    fwd = timeit(lambda: conv(x))
    both = timeit(lambda: torch.autograd.grad(conv(x), [x, conv.weight, conv.bias], g))
    print(json.dumps({"op": name, "forward_us": round(fwd, 1), "backward_us": round(both - fwd, 1)}))
for name, (h, w), (H, W) in (("up_add P4->P3", (50, 84), (100, 167)), ("up_add P5->P4", (25, 42), (50, 84))):
    p = cl(B, 256, h, w).requires_grad_()
    l = cl(B, 256, H, W).requires_grad_()
    f = lambda: F.interpolate(p, size=(H, W), mode="bilinear", align_corners=False) + l      # noqa: E731
    g = torch.randn_like(f())
    fwd = timeit(f)
    both = timeit(lambda: torch.autograd.grad(f(), [p, l], g))
    print(json.dumps({"op": name, "forward_us": round(fwd, 1), "backward_us": round(both - fwd, 1)}))
