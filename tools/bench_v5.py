#!/usr/bin/env python3
"""BASELINE config 5 as a full training step: YOLOv5-l, C = 80, B = 32, 640^2, convolutions under bf16 autocast, loss sums in
fp32 (HIP v5 loss on bf16 heads), fused bf16 BN + LeakyReLU epilogues, fused Adam on fp32 master weights.  Secondary number
(the judged bench is config 2); `--dtype fp32` runs the same network without autocast for comparison."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.model.YOLOV5 import YOLOv5


from objectdetectionpl_amd.optim import Adam as OdplAdam


class Args:
    model_name = "YOLOv5"; cls_criterion = "bce_loss"; coord_criterion = "smooth_l1_loss"; data_module = "synthetic"
    type = "Yolov5l"; lr = 1e-3; weight_decay = 1e-5


def run(dtype="bf16", batch=32, img=640, steps=10, warmup=3, type_="Yolov5l", miopen_find=False, log=None):
    """One measurement on cuda:0 with the class switches as they are set (bench.py's `secondary` leg calls this in-process)."""
    from objectdetectionpl_amd import _lib
    dev = torch.device("cuda:0")
    prev_bench = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(miopen_find)
    try:
        torch.manual_seed(0)
        Args.type = type_
        model = YOLOv5([str(i) for i in range(80)], Args()).to(dev).to(memory_format=torch.channels_last).train()
        model.img_size = img
        opt = OdplAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)             # what configure_optimizers returns (optimizer.py:6-38)
        g = torch.Generator().manual_seed(1)
        x = torch.rand(batch, 3, img, img, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
        n = batch * 7
        wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
        y = torch.cat([torch.arange(batch).repeat_interleave(7).float()[:, None], torch.randint(0, 80, (n, 1), generator=g).float(),
                       torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)

        def step():
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dtype == "bf16"):
                out = model(x)
            d = model.criterion(out, y)
            d["loss"].backward()
            opt.step()
            return d, out

        t0 = time.perf_counter()
        for i in range(warmup):
            d, out = step()
            torch.cuda.synchronize()
            msg = f"[{time.perf_counter() - t0:6.1f}s] config 5 warmup {i + 1}/{warmup}"
            log(msg) if log else print(msg, file=sys.stderr, flush=True)
        PROF = ("yolov5_obj", "yolov5_dense", "headgrad_stream", "headgrad_finish")
        for k in PROF:
            _lib.profile_enable(k, True)
        t1 = time.perf_counter()
        for _ in range(steps):
            d, out = step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t1) / steps
        kern = {}
        for k in PROF:
            _lib.profile_enable(k, False)
            ms = _lib.profile_collect(k)[0]
            if ms:
                kern[k + "_us"] = round(1e3 * sum(ms) / len(ms), 2)
        return {"workload": f"{type_} {img}x{img} bs={batch} training step, convs {dtype}", "img_per_s": round(batch / dt, 1),
                "ms_per_step": round(dt * 1e3, 2), "head_dtype": str(out[0].dtype), "loss": round(float(d["loss"].detach()), 4),
                "miopen_find": bool(miopen_find), "head_contiguous": bool(out[0].is_contiguous()), **kern}
    finally:
        torch.backends.cudnn.benchmark = prev_bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--img", type=int, default=640)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--type", default="Yolov5l")
    ap.add_argument("--miopen-find", action="store_true", help="cudnn.benchmark=True (exhaustive MIOpen search: minutes on a fresh box)")
    ap.add_argument("--no-fused-head", action="store_true", help="A/B: dense d loss/d p instead of the structured Detect backward")
    ap.add_argument("--contiguous-heads", action="store_true", help="A/B: Yolo_Layers copies to [B,na,G,G,no] like the reference")
    ap.add_argument("--no-head-gemm", action="store_true", help="A/B: Detect convolutions on MIOpen")
    ap.add_argument("--no-upsample-cat", action="store_true", help="A/B: neck merges as upsample followed by torch.cat")
    ap.add_argument("--no-bn-of-cat", action="store_true", help="A/B: BottleneckCSP with torch.cat before its BatchNorm (the reference's op sequence)")
    a = ap.parse_args()
    if not a.miopen_find:
        from objectdetectionpl_amd.miopen import stage_user_db
        stage_user_db()                                   # the committed find-db, as bench.py's secondary leg and the Trainer run
    from objectdetectionpl_amd.LightningFunc.losses_dense import MultiScaleRegionLoss_v5
    from objectdetectionpl_amd.LightningFunc.utils.headconv import HeadConv2d
    from objectdetectionpl_amd.model.YOLOV5 import Yolo_Layers
    MultiScaleRegionLoss_v5.fuse_head_backward = not a.no_fused_head
    Yolo_Layers.view_heads = not a.contiguous_heads
    HeadConv2d.enabled = not a.no_head_gemm
    from objectdetectionpl_amd.LightningFunc.utils import bnact as _bnact_mod
    _bnact_mod.bn_of_cat_enabled = not a.no_bn_of_cat
    from objectdetectionpl_amd.LightningFunc.utils import upsample as _up_mod
    _up_mod.cat_enabled = not a.no_upsample_cat
    print(json.dumps(run(a.dtype, a.batch, a.img, a.steps, a.warmup, a.type, a.miopen_find)))


if __name__ == "__main__":
    main()
