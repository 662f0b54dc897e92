#!/usr/bin/env python3
"""1x1 stride-1 convolutions of the YOLOv4 bench (channels-last fp32): MIOpen (find-db solvers) vs the same math as a plain
GEMM on the [N*H*W, Cin] view (rocBLAS / hipBLASLt), forward + backward (dX and dW)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

bench._stage_miopen_db()      # before torch initialises MIOpen
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
shapes = [(16, 64, 64, 208), (16, 128, 64, 104), (16, 128, 128, 104), (16, 256, 128, 52), (16, 256, 256, 52), (16, 512, 256, 26),
          (16, 512, 512, 26), (16, 1024, 512, 13), (16, 2048, 512, 13), (16, 512, 1024, 13)]


def timeit(fn, n=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


print("B,Cin,Cout,HW,conv_fwdbwd_us,gemm_fwdbwd_us,conv_TFs,gemm_TFs")
for B, ci, co, hw in shapes:
    x = torch.randn(B, ci, hw, hw, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).requires_grad_()
    g = torch.randn(B, co, hw, hw, device=dev).contiguous(memory_format=torch.channels_last)

    def conv():
        x.grad = None; w.grad = None
        F.conv2d(x, w).backward(g)

    def gemm():
        x.grad = None; w.grad = None
        z = torch.matmul(x.permute(0, 2, 3, 1).reshape(-1, ci), w.view(co, ci).t())
        z.view(B, hw, hw, co).permute(0, 3, 1, 2).backward(g)

    tc, tg = timeit(conv), timeit(gemm)
    fl = 3 * 2.0 * B * hw * hw * ci * co
    print(f"{B},{ci},{co},{hw},{tc:.1f},{tg:.1f},{fl / tc / 1e6:.1f},{fl / tg / 1e6:.1f}", flush=True)
