#!/usr/bin/env python3
"""A full training step of one of the reference's other model families on synthetic data (no BASELINE config names them; used to put
their convolution problems into the committed MIOpen db and to see that they train): YOLOv2 / YOLOv3 / YOLOv4, fp32 channels-last, the
package's Adam.  `--model YOLOv3 --img 416 --batch 16 --classes 80`."""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.optim import Adam as OdplAdam


def run(model_name, img, batch, classes, steps, warmup, per_image, miopen_find):
    import importlib
    dev = torch.device("cuda:0")
    prev = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(miopen_find)
    try:
        torch.manual_seed(0)
        args = SimpleNamespace(model_name=model_name, cls_criterion="bce_loss", coord_criterion="mse_loss", data_module="synthetic", lr=1e-3, weight_decay=1e-5)
        cls = getattr(importlib.import_module("objectdetectionpl_amd.model." + {"YOLOv2": "YOLOV2", "YOLOv3": "YOLOV3", "YOLOv4": "YOLOV4"}[model_name]), model_name)
        cls.img_size = img
        model = cls([str(i) for i in range(classes)], args).to(dev).to(memory_format=torch.channels_last).train()
        opt = OdplAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(batch, 3, img, img, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        n = batch * per_image
        wh = torch.rand(n, 2, generator=g) * 0.4 + 0.05
        y = torch.cat([torch.arange(batch).repeat_interleave(per_image).float()[:, None], torch.randint(0, classes, (n, 1), generator=g).float(),
                       torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            d = model.training_step((x, y), 0)
            d["loss"].backward()
            opt.step()
            return d

        t0 = time.perf_counter()
        for i in range(warmup):
            d = step()
            torch.cuda.synchronize()
            print(f"[{time.perf_counter() - t0:6.1f}s] {model_name} warmup {i + 1}/{warmup}", file=sys.stderr, flush=True)
        t1 = time.perf_counter()
        for _ in range(steps):
            d = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        return {"workload": f"{model_name} {img}x{img} bs={batch} training step, fp32, {classes} classes", "img_per_s": round(batch / dt, 1),
                "ms_per_step": round(dt * 1e3, 2), "loss": round(float(d["loss"].detach()), 4), "first_step_s": None, "miopen_find": bool(miopen_find)}
    finally:
        torch.backends.cudnn.benchmark = prev


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="YOLOv3", choices=["YOLOv2", "YOLOv3", "YOLOv4"])
    ap.add_argument("--img", type=int, default=416)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--boxes", type=int, default=7)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--miopen-find", action="store_true")
    a = ap.parse_args()
    if not a.miopen_find:
        from objectdetectionpl_amd.miopen import stage_user_db
        stage_user_db()
    out = run(a.model, a.img, a.batch, a.classes, a.steps, a.warmup, a.boxes, a.miopen_find)
    out.pop("first_step_s")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
