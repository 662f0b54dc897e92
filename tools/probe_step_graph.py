#!/usr/bin/env python3
"""Is the whole YOLOv4 training step (forward, HIP loss, backward, fused Adam) capturable as ONE HIP graph, does it replay to the
same numbers as the eager step, and what does it buy?  Also checks the claim of DESIGN.md 4.6 directly: a captured
hipMemsetAsync node (MIOpen's CK split-K kernels zero their output that way)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

bench._stage_miopen_db()
import torch

bench.torch = torch
from objectdetectionpl_amd.model.YOLOV4 import YOLOv4

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True

# ---- 1. hipMemsetAsync inside a captured graph
hip = ctypes.CDLL("libamdhip64.so")
buf = torch.zeros(1 << 20, dtype=torch.float32, device=dev)
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    g = torch.cuda.CUDAGraph()
    buf.fill_(5.0)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        hip.hipMemsetAsync(ctypes.c_void_p(buf.data_ptr()), 0, ctypes.c_size_t(buf.numel() * 4), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        buf.add_(1.0)
    vals = []
    for _ in range(3):
        g.replay()
        torch.cuda.synchronize()
        vals.append((float(buf.min()), float(buf.max())))
print(json.dumps({"memset_node_replays": vals, "expected": "(1.0, 1.0) each time"}), flush=True)


def build(seed=0):
    torch.manual_seed(seed)
    m = YOLOv4([str(i) for i in range(80)], bench.Args()).to(dev).train().to(memory_format=torch.channels_last)
    return m


x, y = bench.synthetic_batch(1, 16, 80, 416, dev)
x = x.contiguous(memory_format=torch.channels_last)
res = {}
for mode in ("eager", "graph"):
    model = build()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=(mode == "graph"))

    def step():
        opt.zero_grad(set_to_none=True)
        d = model.criterion(model(x), y)
        d["loss"].backward()
        opt.step()
        return d["loss"].detach()

    for _ in range(4):
        loss = step()
    torch.cuda.synchronize()
    if mode == "graph":
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            gl = step()
        run = lambda: graph.replay()          # noqa: E731
    else:
        run = step
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for _ in range(K):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    final = float((gl if mode == "graph" else out).item())
    res[mode] = {"ms_per_step": round(dt * 1e3, 3), "img_per_s": round(16 / dt, 1), "loss_after": round(final, 5),
                 "w_sum": float(sum(p.double().sum() for p in model.parameters()))}
print(json.dumps(res))
