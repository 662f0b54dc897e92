#!/usr/bin/env python3
"""One-launch BatchNorm + activation (csrc/odpl_bnfused.hip) against the three-launch path of the same library, per BASELINE
config-2 layer shape: device time of forward and of backward per layer (torch events around 10 back-to-back calls on distinct
tensors, i.e. cold caches for the 44 MB+ working sets), alternating rounds in ONE process."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.utils.bnact import bn_act

dev = torch.device("cuda:0")
# (shape, layers of that shape in YOLOv4 at 416^2)
SHAPES = [((16, 512, 13, 13), 19), ((16, 1024, 13, 13), 7), ((16, 256, 26, 26), 27), ((16, 512, 26, 26), 7), ((16, 128, 52, 52), 23),
          ((16, 256, 52, 52), 5), ((16, 64, 104, 104), 7), ((16, 128, 104, 104), 2)]
ACT = sys.argv[1] if len(sys.argv) > 1 else "mish"
N = 10
rows = []
for shp, layers in SHAPES:
    x = torch.randn(*shp, device=dev).contiguous(memory_format=torch.channels_last)
    w = torch.randn_like(x)
    bn = nn.BatchNorm2d(shp[1]).to(dev).train()
    t = {0: {"fwd": [], "bwd": []}, 1: {"fwd": [], "bwd": []}}
    for rnd in range(4):
        for fused in (1, 0):
            _lib.tune_set("bn_fused", fused)
            for phase in ("fwd", "bwd"):
                xs = [x.clone().requires_grad_() for _ in range(N + 2)]
                ys = [None] * (N + 2)
                if phase == "bwd":
                    ys = [bn_act(v, bn, ACT) for v in xs]
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for i in range(2):
                    if phase == "fwd":
                        ys[i] = bn_act(xs[i], bn, ACT)
                    else:
                        ys[i].backward(w)
                e0.record()
                for i in range(2, N + 2):
                    if phase == "fwd":
                        ys[i] = bn_act(xs[i], bn, ACT)
                    else:
                        ys[i].backward(w)
                e1.record(); torch.cuda.synchronize()
                t[fused][phase].append(e0.elapsed_time(e1) * 1e3 / N)
    med = lambda v: sorted(v)[len(v) // 2]      # noqa: E731
    r = {"shape": shp, "MB": round(x.numel() * 4 / 1e6, 1), "layers": layers, "act": ACT,
         "three_launch_fwd_us": round(med(t[0]["fwd"]), 1), "one_launch_fwd_us": round(med(t[1]["fwd"]), 1),
         "three_launch_bwd_us": round(med(t[0]["bwd"]), 1), "one_launch_bwd_us": round(med(t[1]["bwd"]), 1)}
    r["saved_us_per_step"] = round(layers * (r["three_launch_fwd_us"] - r["one_launch_fwd_us"] + r["three_launch_bwd_us"] - r["one_launch_bwd_us"]), 1)
    rows.append(r)
    print(json.dumps(r), flush=True)
_lib.tune_set("bn_fused", 0)
print(json.dumps({"total_saved_us_per_step": round(sum(r["saved_us_per_step"] for r in rows), 1), "stats": _lib.bn_fused_stats()}))
