#!/usr/bin/env python3
"""The three linear head convolutions of YOLOv4 (1x1, Cin -> 255, bias; model/YOLOV4.py:89-127) on channels-last fp32
activations: MIOpen (find-db solvers, channels-last output) against plain library GEMMs that write the head tensor
NCHW-planar ([B, 255, G*G] = W[255,Cin] @ X_b^T), forward and backward separately.  Decides where the loss reads its
heads from."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: F401
bench._stage_miopen_db()
import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda:0")
big = torch.empty(1 << 27, dtype=torch.float32, device=dev)


def timeit(fn, n=20, cold=True):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(n):
        if cold:
            big.sum()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot * 1e3 / n


print("Cin,G,conv_fwd_us,conv_bwd_us,mm_fwd_us,mm_dx_us,mm_dw_us,copy_to_nchw_us,copy_to_cl_us")
for ci, G in ((256, 52), (512, 26), (1024, 13)):
    B, co = 16, 255
    x = torch.randn(B, ci, G, G, device=dev).contiguous(memory_format=torch.channels_last).requires_grad_()
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).requires_grad_()
    b = torch.zeros(co, device=dev, requires_grad=True)
    g_cl = torch.randn(B, co, G, G, device=dev).contiguous(memory_format=torch.channels_last)
    g_nchw = g_cl.contiguous()
    xm = x.detach().permute(0, 2, 3, 1).reshape(B, G * G, ci)          # [B, G2, Cin] view of the channels-last storage
    wm = w.detach().view(co, ci)

    t_cf = timeit(lambda: F.conv2d(x, w, b))
    y = F.conv2d(x, w, b)

    def cb():
        x.grad = None; w.grad = None; b.grad = None
        y.backward(g_cl, retain_graph=True)
    t_cb = timeit(cb)
    t_mf = timeit(lambda: torch.baddbmm(b.detach().view(1, co, 1), wm.expand(B, co, ci), xm.transpose(1, 2)))
    gm = g_nchw.view(B, co, G * G)
    t_dx = timeit(lambda: torch.matmul(gm.transpose(1, 2), wm))          # [B, G2, Cin]
    t_dw = timeit(lambda: torch.bmm(gm, xm).sum(0))                      # [255, Cin]
    yc = y.detach()
    t_c1 = timeit(lambda: yc.contiguous())
    t_c2 = timeit(lambda: g_nchw.contiguous(memory_format=torch.channels_last))
    print(f"{ci},{G},{t_cf:.1f},{t_cb:.1f},{t_mf:.1f},{t_dx:.1f},{t_dw:.1f},{t_c1:.1f},{t_c2:.1f}", flush=True)
