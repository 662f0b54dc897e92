#!/usr/bin/env python3
"""The loss path alone at BASELINE config 2 (B=16, C=80, 416^2), for `rocprofv3 --kernel-trace --stats`:
  fused   x (channels-last) -> head 1x1 convs -> MultiScaleRegionLoss_v4 -> structured head backward
  generic head tensors -> MultiScaleRegionLoss_v4 -> dense d loss/d head (expand)
MODE=fused|generic|both, N iterations, COLD=1 reads 1 GiB between iterations (evicts L2 + Infinity Cache)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.LightningFunc.losses import MultiScaleRegionLoss_v4
from objectdetectionpl_amd.LightningFunc.utils.headconv import HeadConv2d

A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
M3 = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
dev = torch.device("cuda:0")
N, mode, cold = int(os.environ.get("N", 50)), os.environ.get("MODE", "both"), os.environ.get("COLD", "1") == "1"
B, C = 16, 80
g = torch.Generator().manual_seed(2)
n = B * 7
wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
tgt = torch.cat([torch.arange(B).repeat_interleave(7).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                 torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)
if "HG_ROWS" in os.environ:
    from objectdetectionpl_amd import _lib
    _lib.check(_lib.lib().odpl_head_conv_tune(int(os.environ["HG_ROWS"])), "odpl_head_conv_tune")
crit = MultiScaleRegionLoss_v4(A9, M3, "bce_loss", "smooth_l1_loss", "bce_loss", C, 416)
big = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None
xs = [torch.randn(B, c, G, G, generator=g).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_() for c, G in ((256, 52), (512, 26), (1024, 13))]
convs = [HeadConv2d(c, 255, 1).to(dev) for c in (256, 512, 1024)]
heads = [torch.randn(B, 255, G, G, generator=g).to(dev).requires_grad_() for G in (52, 26, 13)]
for it in range(N):
    if cold:
        big.sum()
    if mode in ("fused", "both"):
        crit([c(x) for c, x in zip(convs, xs)], tgt)["loss"].backward()
    if mode in ("generic", "both"):
        crit(heads, tgt)["loss"].backward()
torch.cuda.synchronize()
print("done")
