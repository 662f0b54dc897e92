#!/usr/bin/env python3
"""The 1x1 convolutions of YOLOv4 (416^2, bs 16) as fp32-MFMA GEMMs (csrc/odpl_gemm1x1.hip) against MIOpen: correctness vs F.conv2d and
device time of forward / backward-data per distinct shape (events around 10 queued calls)."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
bench._stage_miopen_db()
import torch
import torch.nn.functional as F

from objectdetectionpl_amd import _lib

dev = torch.device("cuda:0")
torch.backends.cudnn.benchmark = True
L = _lib.lib()
# (cin, cout, h, layers)
SHAPES = [(128, 64, 208, 1), (64, 64, 208, 3), (128, 128, 104, 1), (256, 256, 52, 1), (512, 512, 26, 1), (1024, 1024, 13, 1), (2048, 512, 13, 1),
          (64, 32, 208, 1), (128, 64, 104, 2), (256, 128, 52, 6), (512, 256, 26, 9), (1024, 512, 13, 8), (64, 64, 104, 3), (128, 128, 52, 9),
          (256, 256, 26, 9), (512, 512, 13, 5), (512, 256, 13, 1), (256, 128, 26, 1)]
B, N = 16, 10


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(N):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / N


tot = {"mi_fwd": 0, "my_fwd": 0, "mi_bwd": 0, "my_bwd": 0}
st = _lib.current_stream(dev)
for ci, co, h, cnt in SHAPES:
    x = torch.randn(B, ci, h, h, device=dev).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(co, ci, 1, 1, device=dev) * 0.05).contiguous(memory_format=torch.channels_last)
    M = B * h * h
    y_ref = F.conv2d(x, w)
    gy = torch.randn_like(y_ref)
    gx_ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    r = {"cin": ci, "cout": co, "h": h, "layers": cnt, "M": M}
    flops = 2.0 * M * ci * co
    r["mi_fwd_us"] = round(timed(lambda: F.conv2d(x, w)), 1)
    r["mi_bwd_us"] = round(timed(lambda: torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])), 1)
    best = {}
    for tile in (64, 128):
        if co % tile == 0 and co >= 64:
            y = torch.empty_like(y_ref)
            rc = L.odpl_conv1x1_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), M, ci, co, tile, st)
            if rc == 0:
                torch.cuda.synchronize()
                err = float((y - y_ref).abs().max() / y_ref.abs().max())
                t = timed(lambda: L.odpl_conv1x1_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), M, ci, co, tile, st))
                r[f"my_fwd{tile}_us"], r[f"fwd{tile}_err"] = round(t, 1), float(f"{err:.2e}")
                best["fwd"] = min(best.get("fwd", 1e9), t)
        if ci % tile == 0 and ci >= 64:
            gx = torch.empty_like(x)
            rc = L.odpl_conv1x1_backward_data(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(gx), M, ci, co, tile, st)
            if rc == 0:
                torch.cuda.synchronize()
                err = float((gx - gx_ref).abs().max() / gx_ref.abs().max())
                t = timed(lambda: L.odpl_conv1x1_backward_data(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(gx), M, ci, co, tile, st))
                r[f"my_bwd{tile}_us"], r[f"bwd{tile}_err"] = round(t, 1), float(f"{err:.2e}")
                best["bwd"] = min(best.get("bwd", 1e9), t)
    r["peak_frac_mi_fwd"] = round(flops / (r["mi_fwd_us"] * 1e-6) / 157.3e12, 2)
    if "fwd" in best:
        r["peak_frac_my_fwd"] = round(flops / (best["fwd"] * 1e-6) / 157.3e12, 2)
    tot["mi_fwd"] += cnt * r["mi_fwd_us"]; tot["mi_bwd"] += cnt * r["mi_bwd_us"]
    tot["my_fwd"] += cnt * min(best.get("fwd", 1e9), r["mi_fwd_us"]); tot["my_bwd"] += cnt * min(best.get("bwd", 1e9), r["mi_bwd_us"])
    print(json.dumps(r), flush=True)
print(json.dumps({k: round(v / 1e3, 3) for k, v in tot.items()}))
