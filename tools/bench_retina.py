#!/usr/bin/env python3
"""Kernel-level measurement of the RetinaNet loss at BASELINE config 4 (B=4, 800x1333 -> 200 700 anchors, C=80)."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes


def main():
    dev = torch.device("cuda:0")
    B, C = int(os.environ.get("B", 4)), 80
    anchors = get_anchor_boxes([1333, 800], dev)
    An = anchors.shape[0]
    g = torch.Generator().manual_seed(4)
    loc = (torch.randn(B, An, 4, generator=g) * 0.3).to(dev).requires_grad_()
    cls = torch.randn(B, An, C, generator=g).to(dev).requires_grad_()
    n = B * 8
    wh = torch.rand(n, 2, generator=g) * 0.45 + 0.05
    tgt = torch.cat([torch.arange(B).repeat_interleave(8).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                     torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1).to(dev)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, 800)     # scalar img_size (losses.py:425)
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
    N = int(os.environ.get("N", 30))
    for cold, nt in ((1, 1), (1, 0), (0, 1), (0, 0)):
        _lib.tune_set("retina_fill_nt", nt)
        for _ in range(3):
            crit((loc, cls), tgt)
        torch.cuda.synchronize()
        for k in ("retina_assign", "retina_main"):
            _lib.profile_enable(k, True)
        t0 = time.perf_counter()
        for _ in range(N):
            if cold:
                big.fill_(1.0)
            out = crit((loc, cls), tgt)
        torch.cuda.synchronize()
        res = {"cold_caches": bool(cold), "nt": nt, "B": B, "An": An, "C": C, "num_pos": int(crit.last_counts[0])}
        for k in ("retina_assign", "retina_main"):
            _lib.profile_enable(k, False)
            ms, _ = _lib.profile_collect(k)
            ms.sort()
            res[k + "_us"] = round(1e3 * ms[len(ms) // 2], 2)
        alg = 4 * B * An * (C + 4) + 4 * B * An          # gradient write + code read
        res["main_algorithmic_bytes"] = alg
        res["main_GBps"] = round(alg / (res["retina_main_us"] * 1e-6) / 1e9, 1)
        res["main_frac_of_8TBps"] = round(res["main_GBps"] / 8000.0, 4)
        res["survey_8d_bytes"] = 2 * 4 * B * An * (C + 4) + 16 * An
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
