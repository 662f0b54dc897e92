#!/usr/bin/env python3
"""Per-layer MFMA utilisation of the YOLOv4 convolutions on MIOpen (fp32, channels-last, cudnn.benchmark): every distinct
(Cin, Cout, k, stride, H) of the network at 416^2, bs 16 -- forward, backward-data and backward-weights timed separately with
events around 10 calls (device-bound: the calls are queued back to back), against 2*MACs and the 157.3 TFLOP/s fp32 MFMA peak."""
import collections
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench._stage_miopen_db()
import torch
import torch.nn as nn
bench.torch = torch
from objectdetectionpl_amd.model.YOLOV4 import YOLOv4

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
model = YOLOv4([str(i) for i in range(80)], bench.Args()).eval()
shapes = collections.Counter()


def hook(m, inp, out):
    x = inp[0]
    shapes[(m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], x.shape[2])] += 1


for m in model.modules():
    if isinstance(m, nn.Conv2d):
        m.register_forward_hook(hook)
with torch.no_grad():
    model(torch.rand(1, 3, 416, 416))
B, N = 16, 10
rows = []
tot = {"fwd": 0.0, "bwd": 0.0, "wrw": 0.0}
for (ci, co, k, s, h), cnt in sorted(shapes.items(), key=lambda kv: -kv[0][0] * kv[0][1] * kv[0][2] ** 2 * (kv[0][4] // kv[0][3]) ** 2):
    x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, k, k, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    pad = k // 2
    y = torch.nn.functional.conv2d(x, w, None, s, pad)
    gy = torch.randn_like(y)
    ho = y.shape[2]
    flops = 2.0 * B * ho * ho * co * ci * k * k
    t = {}
    for name, fn in (("fwd", lambda: torch.nn.functional.conv2d(x, w, None, s, pad)),
                     ("bwd", lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [True, False, False])),
                     ("wrw", lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [s, s], [pad, pad], [1, 1], False, [0, 0], 1, [False, True, False]))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(N):
            fn()
        e1.record(); torch.cuda.synchronize()
        t[name] = e0.elapsed_time(e1) * 1e3 / N
        tot[name] += cnt * t[name]
    r = {"cin": ci, "cout": co, "k": k, "stride": s, "h_in": h, "layers": cnt, "gflop": round(flops / 1e9, 2),
         **{f"{n}_us": round(v, 1) for n, v in t.items()}, **{f"{n}_frac": round(flops / (v * 1e-6) / 157.3e12, 3) for n, v in t.items()}}
    rows.append(r)
    print(json.dumps(r), flush=True)
print(json.dumps({"total_ms_per_step": {k: round(v / 1e3, 2) for k, v in tot.items()}, "sum_ms": round(sum(tot.values()) / 1e3, 2)}))
