#!/usr/bin/env python3
"""GPU tuning harness for the YOLO loss kernels at the BASELINE config-2 shape (B=16, C=80, 416^2).
Per setting: N calls of the criterion (fwd+bwd buffers), per-kernel time from the in-library HIP events."""
import itertools
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.losses import MultiScaleRegionLoss_v4

A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
M = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(2)
    B, C = 16, 80
    heads = [torch.randn(B, 255, G, G, generator=g).to(dev).requires_grad_() for G in (52, 26, 13)]
    n = B * 7
    wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
    tgt = torch.cat([torch.arange(B).repeat_interleave(7).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                     torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)
    crit = MultiScaleRegionLoss_v4(A9, M, "bce_loss", "smooth_l1_loss", "bce_loss", C, 416)
    names = ("yolo_assign", "yolo_loss", "yolo_finalize", "yolo_expand")
    N = int(os.environ.get("N", "100"))
    settings = list(itertools.product([512, 1024, 2048, 4096, 8192, 16384], [0, 1]))
    cold = os.environ.get("COLD", "1") == "1"
    big = torch.empty(1 << 28, dtype=torch.float32, device=dev) if cold else None     # 1 GiB: evicts L2 + Infinity Cache
    print("fill_f4,nt," + ",".join(f"{k}_us" for k in names) + ",call_wall_us")
    for f4, nt in settings:
        _lib.tune_set("yolo_fill_f4_per_block", f4)
        _lib.tune_set("yolo_fill_nt", nt)
        for _ in range(5):
            crit(heads, tgt)["loss"].backward()
        torch.cuda.synchronize()
        for k in names:
            _lib.profile_enable(k, True)
        t0 = time.perf_counter()
        for _ in range(N):
            if cold:
                big.fill_(1.0)
            crit(heads, tgt)["loss"].backward()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / N * 1e6
        row = []
        for k in names:
            _lib.profile_enable(k, False)
            ms, raw = _lib.profile_collect(k)
            ms.sort()
            row.append(f"{1e3 * ms[len(ms) // 2]:.2f}")
        print(f"{f4},{nt}," + ",".join(row) + f",{wall:.1f}", flush=True)


if __name__ == "__main__":
    main()
