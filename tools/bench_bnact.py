#!/usr/bin/env python3
"""GPU time of fused BN+Mish (libodpl_hip) vs nn.BatchNorm2d + F.mish (MIOpen + ATen) per layer shape, fwd+bwd."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn
import torch.nn.functional as F

from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act

dev = torch.device("cuda:0")
shapes = [(16, 32, 416, 416), (16, 64, 208, 208), (16, 128, 104, 104), (16, 256, 52, 52), (16, 512, 26, 26), (16, 1024, 13, 13)]
print("shape,MB,torch_fwd_us,fused_fwd_us,torch_bwd_us,fused_bwd_us")
for shp in shapes:
    x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn_like(x)
    bn = nn.BatchNorm2d(shp[1]).to(dev).train()
    res = []
    for fused in (False, True):
        def fwd(xx):
            return bn_act(xx, bn, "mish") if fused else F.mish(bn(xx))
        ts = []
        for phase in ("fwd", "bwd"):
            xs = [x.clone().requires_grad_() for _ in range(12)]
            ys = [None] * 12
            if phase == "bwd":
                ys = [fwd(v) for v in xs]
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(2, 12):
                if phase == "fwd":
                    ys[i] = fwd(xs[i])
                else:
                    ys[i].backward(w)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)      # us per call (10 calls)
        res.append(ts)
    mb = x.numel() * 4 / 1e6
    print(f"{shp},{mb:.0f},{res[0][0]:.0f},{res[1][0]:.0f},{res[0][1]:.0f},{res[1][1]:.0f}", flush=True)
