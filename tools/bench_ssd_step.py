#!/usr/bin/env python3
"""SSD300-VGG16 as a full training step (no BASELINE config names SSD; this is the A/B for the structured backward of its head pairs):
B = 32, 300 x 300, 21 classes, --boxes per image, fp32 channels-last, the package's Adam.  Default: the reference's plain head modules,
SSDLoss hands autograd the dense d loss/d preds (MIOpen backward of the twelve head convolutions).  `--structured-head-backward`: the
head pairs hosted (utils/ssdhead.py), gradient rows on positives + mined negatives only, odpl_retina_heads_backward per source."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss
from objectdetectionpl_amd.LightningFunc.utils import ssdhead
from objectdetectionpl_amd.model.SSD import SSD
from objectdetectionpl_amd.optim import Adam as OdplAdam


def run(batch=32, steps=10, warmup=3, per_image=8, classes=21, miopen_find=False):
    dev = torch.device("cuda:0")
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(miopen_find)
    try:
        torch.manual_seed(0)
        args = SimpleNamespace(model_name="SSD", cls_criterion="ce_loss", coord_criterion="smooth_l1_loss", data_module="synthetic", lr=1e-3, weight_decay=1e-5)
        model = SSD([str(i) for i in range(classes)], args).to(dev).to(memory_format=torch.channels_last).train()
        opt = OdplAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(batch, 3, 300, 300, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        n = batch * per_image
        wh = torch.rand(n, 2, generator=g) * 0.45 + 0.05
        y = torch.cat([torch.arange(batch).repeat_interleave(per_image).float()[:, None], torch.randint(1, classes, (n, 1), generator=g).float(),
                       torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            d = model.criterion(model(x), y)
            d["loss"].backward()
            opt.step()
            return d

        t0 = time.perf_counter()
        for i in range(warmup):
            d = step()
            torch.cuda.synchronize()
            print(f"[{time.perf_counter() - t0:6.1f}s] SSD warmup {i + 1}/{warmup}", file=sys.stderr, flush=True)
        t1 = time.perf_counter()
        for _ in range(steps):
            d = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        return {"workload": f"SSD300-VGG16 {classes} classes bs={batch} training step, fp32", "img_per_s": round(batch / dt, 1), "ms_per_step": round(dt * 1e3, 2),
                "loss": round(float(d["loss"].detach()), 4), "hosted_heads": bool(ssdhead.enabled), "structured_head_backward": bool(ssdhead.enabled and SSDLoss.fuse_head_backward and (SSDLoss.structured_max_boxes_per_image is None or per_image <= SSDLoss.structured_max_boxes_per_image)), "boxes_per_image": per_image, "miopen_find": bool(miopen_find)}
    finally:
        torch.backends.cudnn.benchmark = prev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--miopen-find", action="store_true")
    ap.add_argument("--structured-head-backward", action="store_true", help="always (default: SSDLoss decides from the boxes per image)")
    ap.add_argument("--dense-head-backward", action="store_true", help="always")
    ap.add_argument("--plain-heads", action="store_true", help="the reference's nn.Conv2d modules + cat, no hosting")
    ap.add_argument("--boxes", type=int, default=8)
    a = ap.parse_args()
    if not a.miopen_find:
        from objectdetectionpl_amd.miopen import stage_user_db
        stage_user_db()
    ssdhead.enabled = not a.plain_heads
    if a.structured_head_backward:
        SSDLoss.structured_max_boxes_per_image = None
    if a.dense_head_backward:
        SSDLoss.fuse_head_backward = False
    print(json.dumps(run(a.batch, a.steps, a.warmup, per_image=a.boxes, miopen_find=a.miopen_find)))


if __name__ == "__main__":
    main()
