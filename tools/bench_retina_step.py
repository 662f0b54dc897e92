#!/usr/bin/env python3
"""BASELINE config 4 as a full training step (secondary number; the judged bench is config 2): RetinaNet-R50-FPN, C = 80,
B = 4, 800 x 1333 input (x[4, 3, 800, 1333]) -> 200 700 anchors, focal + SmoothL1 through the HIP RetinaNet loss with the
per-axis target scale (width 1333, height 800), fused Adam, fp32, channels-last.  MIOpen immediate mode unless --miopen-find."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
from objectdetectionpl_amd.model.RetinaNet import RetinaNet


from objectdetectionpl_amd.optim import Adam as OdplAdam


class Args:
    model_name = "RetinaNet"; cls_criterion = "focal_loss"; coord_criterion = "smooth_l1_loss"; data_module = "synthetic"
    lr = 1e-3; weight_decay = 1e-5


def run(batch=4, height=800, width=1333, steps=10, warmup=3, miopen_find=False, gt_per_image=8, wh_range=(0.05, 0.5), log=None):
    """One measurement on cuda:0 with the class switches as they are set; returns the result dict (bench.py's `secondary` leg calls
    this in-process after the judged workload)."""
    a = argparse.Namespace(batch=batch, height=height, width=width, steps=steps, warmup=warmup, miopen_find=miopen_find,
                           gt_per_image=gt_per_image, wh_range=tuple(wh_range))
    from objectdetectionpl_amd.LightningFunc.utils import retinahead, sparseconv
    dev = torch.device("cuda:0")
    prev_bench = torch.backends.cudnn.benchmark
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    try:
        return _measure(a, dev, retinahead, sparseconv, log)
    finally:
        torch.backends.cudnn.benchmark = prev_bench


def _measure(a, dev, retinahead, sparseconv, log):
    torch.manual_seed(0)
    model = RetinaNet([str(i) for i in range(80)], Args()).to(dev).to(memory_format=torch.channels_last).train()
    anchors = get_anchor_boxes(torch.Tensor([a.width, a.height]), dev)              # RetinaUtils.get_anchor_boxes takes (w, h)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", 80, (a.width, a.height))
    opt = OdplAdam(model.parameters(), lr=1e-3, weight_decay=1e-5)             # what configure_optimizers returns (optimizer.py:6-38)
    g = torch.Generator().manual_seed(1)
    x = torch.rand(a.batch, 3, a.height, a.width, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    n = a.batch * a.gt_per_image
    wh = torch.rand(n, 2, generator=g) * (a.wh_range[1] - a.wh_range[0]) + a.wh_range[0]
    y = torch.cat([torch.arange(a.batch).repeat_interleave(a.gt_per_image).float()[:, None], torch.randint(0, 80, (n, 1), generator=g).float(),
                   torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1).to(dev)

    model.iou_boxes, model.criterion = anchors, crit      # the 1333 x 800 table and its criterion (the ctor builds img_size^2 ones)

    def step():
        opt.zero_grad(set_to_none=True)
        d = model.training_step((x, y), 0)                # step.py:14-25
        d["loss"].backward()
        opt.step()
        return d

    t0 = time.perf_counter()
    for i in range(a.warmup):
        d = step()
        torch.cuda.synchronize()
        msg = f"[{time.perf_counter() - t0:6.1f}s] config 4 warmup {i + 1}/{a.warmup}"
        log(msg) if log else print(msg, file=sys.stderr, flush=True)
    PROF = ("retina_assign", "retina_main")
    for k in PROF:
        _lib.profile_enable(k, True)
    t1 = time.perf_counter()
    for _ in range(a.steps):
        d = step()
    t_enq = (time.perf_counter() - t1) / a.steps                                  # host time to enqueue a step (== the step when host-bound)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t1) / a.steps
    kern = {}
    for k in PROF:
        _lib.profile_enable(k, False)
        ms = _lib.profile_collect(k)[0]
        if ms:
            kern[k + "_us"] = round(1e3 * sum(ms) / len(ms), 2)
    assert anchors.shape[0] == 200700 or (a.height, a.width) != (800, 1333)
    dense_fwd = not RetinaNet.loss_driven_step
    hosted, fused_bwd, sparse_bwd = retinahead.enabled, RetinaNetLoss.fuse_head_backward, sparseconv.enabled
    return {"workload": f"RetinaNet-R50-FPN {a.height}x{a.width} bs={a.batch} training step, fp32, C=80, {anchors.shape[0]} anchors",
            "img_per_s": round(a.batch / dt, 2), "ms_per_step": round(dt * 1e3, 2), "host_enqueue_ms_per_step": round(t_enq * 1e3, 2), "loss": round(float(d["loss"].detach()), 4),
            "num_pos": int(crit.last_counts[0]), "gt_per_image": a.gt_per_image, "wh_range": list(a.wh_range), "miopen_find": bool(a.miopen_find),
            "hosted_heads": bool(hosted), "structured_head_backward": bool(hosted and fused_bwd),
            "sparse_tower_backward": bool(sparse_bwd),
            "loss_driven_forward": bool(not dense_fwd and hosted and fused_bwd and sparse_bwd),
            "topdown_on_pixels": bool(RetinaNet.topdown_on_pixels and not dense_fwd and hosted and fused_bwd and sparse_bwd),
            **kern}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--height", type=int, default=800)
    ap.add_argument("--width", type=int, default=1333)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--miopen-find", action="store_true")
    ap.add_argument("--no-hosted-heads", action="store_true", help="A/B: the final head convolutions as plain nn.Conv2d modules + cat (round 2)")
    ap.add_argument("--dense-head-backward", action="store_true",
                    help="A/B: hosted final convolutions, but the criterion hands autograd the dense d loss/d preds (aten convolution_backward)")
    ap.add_argument("--dense-forward", action="store_true",
                    help="A/B: training_step = forward + criterion on all 200 700 anchors (round 3 before the loss-driven step); default: the "
                         "assignment runs first and the heads are evaluated on the pixels RetinaNetLoss reads (model/RetinaNet.py:forward_and_loss)")
    ap.add_argument("--dense-topdown", action="store_true", help="A/B: the FPN's top-down path (laterals, merges, smoothing) on every pixel")
    ap.add_argument("--dense-tower-backward", action="store_true", help="A/B: the head towers' 3x3 convolutions keep MIOpen's backward on every level")
    ap.add_argument("--gt-per-image", type=int, default=8)
    ap.add_argument("--wh-range", type=float, nargs=2, default=(0.05, 0.5), metavar=("LO", "HI"),
                    help="box width / height as a share of the image, uniform in [LO, HI] (SURVEY config 4: 0.05 0.5; smaller boxes put "
                         "positives on the large FPN levels, where the loss-driven step has more pixels to evaluate)")
    ap.add_argument("--min-pixels", type=int, default=None, help="sparseconv.min_pixels: levels with fewer B*H*W pixels keep the library route")
    a = ap.parse_args()
    if not a.miopen_find:
        from objectdetectionpl_amd.miopen import stage_user_db
        stage_user_db()                                   # the committed find-db, as bench.py's secondary leg and the Trainer run
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv
    sparseconv.enabled = not a.dense_tower_backward
    if a.min_pixels is not None:
        sparseconv.min_pixels = a.min_pixels
    from objectdetectionpl_amd.LightningFunc.utils import retinahead
    retinahead.enabled = not a.no_hosted_heads
    RetinaNetLoss.fuse_head_backward = not a.dense_head_backward
    RetinaNet.loss_driven_step = not a.dense_forward
    RetinaNet.topdown_on_pixels = not a.dense_topdown
    print(json.dumps(run(a.batch, a.height, a.width, a.steps, a.warmup, a.miopen_find, a.gt_per_image, a.wh_range)))


if __name__ == "__main__":
    main()
