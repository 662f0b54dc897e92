#!/usr/bin/env python3
"""Kernel-level roofline table for every hot-path family at the BASELINE shapes.  Cold caches: a 1 GiB buffer is
READ between calls (clean eviction of L2 + Infinity Cache; EVICT=write fills it instead, which leaves 256 MB of
dirty lines draining under the timed kernel and roughly halves read bandwidth).  HIP-event timing by the library,
three events per launch.  One JSON line per kernel."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib
from objectdetectionpl_amd.LightningFunc import nms
from objectdetectionpl_amd.LightningFunc.losses import MultiScaleRegionLoss_v4
from objectdetectionpl_amd.LightningFunc.losses_dense import MultiScaleRegionLoss_v5, RetinaNetLoss, focal_loss
from objectdetectionpl_amd.LightningFunc.utils import YoloV4Utils
from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes

dev = torch.device("cuda:0")
big = torch.empty(1 << 28, dtype=torch.float32, device=dev)
N = int(os.environ.get("N", 20))
A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
A18 = [v for a in A9 for v in a]
M3 = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
V5A = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def targets(g, B, per, C, lo=0.02, hi=0.5):
    n = B * per
    wh = torch.rand(n, 2, generator=g) * (hi - lo) + lo
    return torch.cat([torch.arange(B).repeat_interleave(per).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                      torch.rand(n, 2, generator=g) * (1 - wh), wh], 1).to(dev)


def measure(name, kernels, fn, alg_bytes, note):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    for k in kernels:
        _lib.profile_enable(k, True)
    for _ in range(N):
        if os.environ.get("EVICT", "read") == "write":
            big.fill_(1.0)
        else:
            big.sum()
        fn()
    torch.cuda.synchronize()
    for k in kernels:
        _lib.profile_enable(k, False)
        ms, _ = _lib.profile_collect(k)
        if not ms:                      # (a launch this route no longer makes, e.g. yolo_assign when the assignment runs inside yolo_loss)
            continue
        ms.sort()
        us = 1e3 * ms[len(ms) // 2]
        b = alg_bytes.get(k)
        rec = {"workload": name, "kernel": k, "median_us": round(us, 2), "evict": os.environ.get("EVICT", "read")}
        if b:
            rec.update(algorithmic_bytes=b, GBps=round(b / (us * 1e-6) / 1e9, 1), frac_of_8TBps=round(b / (us * 1e-6) / 8e12, 4))
        rec["note"] = note.get(k, "")
        print(json.dumps(rec), flush=True)


def main():
    g = torch.Generator().manual_seed(3)
    # ---- config 2: YOLOv4 loss, NMS, decode
    B, C = 16, 80
    heads = [torch.randn(B, 255, G, G, generator=g).to(dev).requires_grad_() for G in (52, 26, 13)]
    tgt = targets(g, B, 7, C)
    crit = MultiScaleRegionLoss_v4(A9, M3, "bce_loss", "smooth_l1_loss", "bce_loss", C, 416)
    cells = sum(B * 3 * G * G for G in (52, 26, 13))
    measure("config2 YOLOv4 loss fwd+bwd (B=16, C=80, 416^2)", ["yolo_assign", "yolo_loss", "yolo_finalize", "yolo_expand"],
            lambda: crit(heads, tgt)["loss"].backward(), {"yolo_expand": 4 * cells * 85 + 4 * cells, "yolo_loss": 16 * cells},
            {"yolo_expand": "dense grad write + compact conf-grad read", "yolo_loss": "conf logit + state read, state reset, compact grad write", "yolo_assign": "latency bound",
             "yolo_finalize": "latency bound"})
    hn = []
    for G in (52, 26, 13):       # valid geometry so that the NMS does real work
        h = torch.randn(B, 3, 85, G, G, generator=g)
        h[:, :, 0:2] = torch.rand(B, 3, 2, 1, 1, generator=g) * 300 + 50 + torch.randn(B, 3, 2, G, G, generator=g) * 40
        h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 80 + 20
        h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.8 - 1.0
        h[:, :, 5:] = torch.rand(B, 3, 80, G, G, generator=g)
        hn.append(h.view(B, 255, G, G).contiguous().to(dev))
    kept = sum(o.shape[0] for o in nms.yolo_non_max_suppression(hn) if o is not None)
    measure(f"config2 YOLO weighted-merge NMS (10647 cand/img, {kept} kept over 16 img)", ["yolo_nms_prep", "yolo_nms_image"],
            lambda: nms.yolo_non_max_suppression(hn), {"yolo_nms_prep": 4 * cells * 85 + cells * 40},
            {"yolo_nms_prep": "head read + SoA write", "yolo_nms_image": "[r4] bucket (1 WG/image) + class (1 WG per image and class) + order (1 WG/image): three launches"})
    measure(f"config2 YOLO weighted-merge NMS, one workgroup per image (round 3's kernel; A/B flag)", ["yolo_nms_image"],
            lambda: nms.yolo_non_max_suppression(hn, one_workgroup=True), {},
            {"yolo_nms_image": "1 WG/image: LDS bitonic sort of every valid candidate + one wave per class"})
    measure("config2 YOLOv4 inference decode", ["yolo_decode"],
            lambda: YoloV4Utils.decode_heads([h.detach() for h in heads], C, A18, M3, [8, 16, 32]),
            {"yolo_decode": 4 * cells * 85 + 4 * cells * 84}, {"yolo_decode": "head read + boxes/confs write"})
    del heads, hn
    # ---- config 4: RetinaNet loss + greedy NMS
    B, C = 4, 80
    anchors = get_anchor_boxes([1333, 800], dev)
    An = anchors.shape[0]
    loc = (torch.randn(B, An, 4, generator=g) * 0.3).to(dev).requires_grad_()
    cls = torch.randn(B, An, C, generator=g).to(dev).requires_grad_()
    tg = targets(g, B, 8, C, 0.05, 0.5)
    rc = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, (1333, 800))
    measure(f"config4 RetinaNet loss fwd+bwd (B=4, {An} anchors, C=80)", ["retina_assign", "retina_main"],
            lambda: rc((loc, cls), tg)["loss"].backward(), {"retina_main": 4 * B * An * 84 + 4 * B * An, "retina_assign": 16 * An + 4 * B * An},
            {"retina_main": "grad write + code read", "retina_assign": "anchors read + code write; IoU compute"})
    cls_n = (cls.detach() - 3.0)
    measure("config4 Retina greedy top-100 NMS", ["greedy_nms_prep", "greedy_nms_image"],
            lambda: nms.greedy_non_max_suppression((loc.detach(), cls_n), anchors),
            {"greedy_nms_prep": 4 * B * An * C + 8 * B * An}, {"greedy_nms_prep": "class rows read (LDS staged) + score/label write",
                                                                 "greedy_nms_image": "[r4] scan + scatter + final (1 WG/image: short top-k, ballot overlap matrix, scalar greedy loop)"})
    del loc, cls, cls_n
    # ---- config 5: YOLOv5-l loss; heads as the reference hands them (contiguous [B,na,G,G,no] = "interleaved") and as the
    # permuted VIEW of the Detect convolution's NCHW output ("planar": what Yolo_Layers returns here)
    B, C = 32, 80
    t5 = targets(g, B, 7, C)
    c5 = MultiScaleRegionLoss_v5(V5A, None, None, None, None, C, 640)
    cells5 = sum(B * 3 * G * G for G in (80, 40, 20))
    base = [torch.randn(B, 3 * 85, G, G, generator=g).to(dev) for G in (80, 40, 20)]

    def views(dtype, planar):
        out = []
        for t in base:
            t = t.to(dtype)
            v = t.view(B, 3, 85, t.shape[2], t.shape[3]).permute(0, 1, 3, 4, 2)
            out.append((v if planar else v.contiguous()).detach().requires_grad_())
        return out
    for dtype, es, tag in ((torch.float32, 4, "fp32"), (torch.bfloat16, 2, "bf16 heads -- the BASELINE dtype")):
        for planar in (False, True):
            hs = views(dtype, planar)
            measure(f"config5 YOLOv5 loss fwd+bwd (B=32, C=80, 640^2, {tag}, {'planar view' if planar else 'interleaved'})",
                    ["yolov5_obj", "yolov5_dense"], lambda: c5(hs, t5)["loss"].backward(),
                    {"yolov5_dense": es * cells5 * 85 + 4 * cells5, "yolov5_obj": es * cells5 + 12 * cells5},
                    {"yolov5_dense": "v5_expand_kernel: dense grad write + compact objectness-grad read",
                     "yolov5_obj": "objectness logit + owner map read, owner reset, compact grad write"})
            del hs
    del base
    caller_side(g)


def timed_events(name, fn, alg_bytes, note, n=None):
    """Caller-side kernels without a library profiling slot: torch events around the call (includes ~2 us of launch overhead)."""
    n = n or N
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        big.sum()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    us = ts[len(ts) // 2]
    print(json.dumps({"workload": name, "kernel": "(events around the call)", "median_us": round(us, 2), "evict": "read", "algorithmic_bytes": alg_bytes,
                      "GBps": round(alg_bytes / (us * 1e-6) / 1e9, 1), "frac_of_8TBps": round(alg_bytes / (us * 1e-6) / 8e12, 4), "note": note}), flush=True)


def caller_side(g):
    """the data movement around the heads, at the BASELINE shapes, one launch each."""
    from objectdetectionpl_amd.LightningFunc.utils.bnact import _BiasActFn
    from objectdetectionpl_amd.LightningFunc.utils.s2d import focus
    from objectdetectionpl_amd.LightningFunc.utils.spp import spp_cat
    from objectdetectionpl_amd.LightningFunc.utils.upsample import upsample_nearest
    cl = torch.channels_last
    for tag, shape, dt, es in (("config2 13x13x512 fp32", (16, 512, 13, 13), torch.float32, 4), ("config5 20x20x512 bf16", (32, 512, 20, 20), torch.bfloat16, 2)):
        x = torch.randn(*shape, generator=g).to(dev).to(dt).contiguous(memory_format=cl).requires_grad_()
        nb = x.numel() * es
        order = "v4" if dt == torch.float32 else "v5"
        timed_events(f"SPP forward ({tag})", lambda: spp_cat(x, order), 5 * nb, "x read, [m13|m9|m5|x] written")
        y = spp_cat(x, order)
        gy = torch.randn_like(y)
        timed_events(f"SPP backward ({tag})", lambda: torch.autograd.grad(y, x, gy, retain_graph=True), 6 * nb, "x + 4 gradient blocks read, grad_x written")
    for tag, shape, dt, es in (("config2 26x26x128 fp32", (16, 128, 26, 26), torch.float32, 4), ("config5 40x40x256 bf16", (32, 256, 40, 40), torch.bfloat16, 2)):
        x = torch.randn(*shape, generator=g).to(dev).to(dt).contiguous(memory_format=cl).requires_grad_()
        nb = x.numel() * es
        size = (2 * shape[2], 2 * shape[3])
        timed_events(f"nearest 2x upsample forward ({tag})", lambda: upsample_nearest(x, size), 5 * nb, "1 read + 4 writes")
        y = upsample_nearest(x, size)
        gy = torch.randn_like(y)
        timed_events(f"nearest 2x upsample backward ({tag})", lambda: torch.autograd.grad(y, x, gy, retain_graph=True), 5 * nb, "4 reads + 1 write")
    x = torch.rand(32, 3, 640, 640, generator=g).to(dev).contiguous(memory_format=cl)
    timed_events("Focus forward (config5 32x3x640x640 fp32, channels-last)", lambda: focus(x), 2 * x.numel() * 4, "pure permutation")
    z = torch.randn(4, 256, 100, 167, generator=g).to(dev).contiguous(memory_format=cl)
    b = torch.randn(256, generator=g).to(dev)
    nb = z.numel() * 4
    timed_events("conv bias + ReLU epilogue forward (config4 P3 head unit)", lambda: _BiasActFn.apply(z, b, 3), 2 * nb, "in place: 1 read + 1 write")
    y = _BiasActFn.apply(z.clone(), b, 3)
    gy, gx, gb = torch.randn_like(y), torch.empty_like(y), torch.empty(256, device=dev)
    L = _lib.lib()
    R = z.shape[0] * z.shape[2] * z.shape[3]
    need = L.odpl_bn_act_workspace_bytes(R, 256) + 4 * 256
    ws = torch.empty(need, dtype=torch.uint8, device=dev)
    timed_events("conv bias + ReLU epilogue backward (config4 P3 head unit)",
                 lambda: _lib.check(L.odpl_bias_act_backward(_lib.ptr(y), _lib.ptr(gy), R, 256, 3, _lib.ptr(gx), _lib.ptr(gb), _lib.ptr(ws), need,
                                                             _lib.current_stream(dev)), "odpl_bias_act_backward"),
                 3 * nb, "gy + y read, grad written, bias-gradient partials + finalize launch")


if __name__ == "__main__":
    main()

