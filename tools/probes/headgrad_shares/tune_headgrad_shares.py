#!/usr/bin/env python3
"""Tile-height sweep of `headgrad_stream_kernel` at BASELINE config 2 (heads of YOLOv4: Cin 256/512/1024 at G 52/26/13, B = 16),
cold caches, library event timing.  The 1 GiB eviction read sits BETWEEN forward and backward: in a training step the X of the
first head was read by its forward convolution dozens of layers before the backward starts; evicting only between iterations
leaves all of X in the 256 MB Infinity Cache and flatters the kernel by ~10 %.  ROWS="16 32 64" overrides the list; 0 = the auto height."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.losses import MultiScaleRegionLoss_v4
from objectdetectionpl_amd.LightningFunc.utils.headconv import HeadConv2d

A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
M3 = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
dev = torch.device("cuda:0")
B, C, N = 16, 80, int(os.environ.get("N", 25))
g = torch.Generator().manual_seed(2)
n = B * 7
wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
tgt = torch.cat([torch.arange(B).repeat_interleave(7).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                 torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)
crit = MultiScaleRegionLoss_v4(A9, M3, "bce_loss", "smooth_l1_loss", "bce_loss", C, 416)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
xs = [torch.randn(B, c, G, G, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_() for c, G in ((256, 52), (512, 26), (1024, 13))]
convs = [HeadConv2d(c, 255, 1).to(dev) for c in (256, 512, 1024)]
alg = sum(2 * 4 * B * G * G * c for c, G in ((256, 52), (512, 26), (1024, 13))) + 4 * sum(B * 3 * G * G for G in (52, 26, 13))
# ROWS entries: a positive tile height, or s<workgroups> for the equal-share mode with that many workgroups (e.g. s512)
for rows in os.environ.get("ROWS", "32 s256 s384 s512 s640 s768 s1024 s1536").split():
    if rows.startswith("s"):
        _lib.check(_lib.lib().odpl_head_conv_tune(0), "odpl_head_conv_tune")
        _lib.check(_lib.lib().odpl_head_conv_tune_shares(int(rows[1:])), "odpl_head_conv_tune_shares")
    else:
        _lib.check(_lib.lib().odpl_head_conv_tune(int(rows)), "odpl_head_conv_tune")
    for _ in range(3):
        crit([c(x) for c, x in zip(convs, xs)], tgt)["loss"].backward()
    for k in ("headgrad_stream", "headgrad_finish"):
        _lib.profile_enable(k, True)
    for _ in range(N):
        loss = crit([c(x) for c, x in zip(convs, xs)], tgt)["loss"]
        big.sum()
        loss.backward()
    torch.cuda.synchronize()
    rec = {"rows_per_tile": rows}
    for k in ("headgrad_stream", "headgrad_finish"):
        _lib.profile_enable(k, False)
        ms = sorted(_lib.profile_collect(k)[0])
        rec[k + "_us"] = round(1e3 * ms[len(ms) // 2], 2)
    rec["stream_frac_of_8TBps"] = round(alg / (rec["headgrad_stream_us"] * 1e-6) / 8e12, 4)
    print(json.dumps(rec), flush=True)
