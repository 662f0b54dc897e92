#!/usr/bin/env python3
"""Evaluation-side throughput at the bench shape: YOLOv4 416^2, bs 16 -- forward (eval mode), validation_step
(forward + loss, no grad) and test_step (forward + weighted-merge NMS + get_yolo_statistics)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as B

B._stage_miopen_db()        # before torch initialises MIOpen
import torch

B.torch = torch            # bench.py imports torch lazily (its --gpus parent must never touch the GPU)
from objectdetectionpl_amd.model.YOLOV4 import YOLOv4


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model = YOLOv4([str(i) for i in range(80)], B.Args()).to(dev).to(memory_format=torch.channels_last).eval()
    x, y = B.synthetic_batch(1, 16, 80, 416, dev)
    x = x.contiguous(memory_format=torch.channels_last)

    def t(fn, n=20, w=4):
        for _ in range(w):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    with torch.no_grad():
        f = t(lambda: model(x))
    v = t(lambda: model.validation_step((x, y), 0))
    te = t(lambda: model.test_step((x, y.clone()), 0))
    with torch.no_grad():
        out = model(x)
        nms = t(lambda: model.non_max_suppression(out))
        yy = y.clone(); yy[:, 2:] *= 416
        st = t(lambda: model.get_yolo_statistics(out, yy))
        kept = sum(0 if o is None else o.shape[0] for o in model.non_max_suppression(out))
    print(f"nms_ms={nms:.2f} (kept {kept} boxes over 16 images)  get_yolo_statistics_ms={st:.2f}")
    # the same forward captured in a HIP graph (serving path: no per-launch host cost)
    gfwd = None
    try:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(3):
                model(x)
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(graph):
            static_out = model(x)
        gfwd = t(graph.replay)
        with torch.no_grad():
            ref = model(x)
        diff = max(float((a - b).abs().max()) for a, b in zip(static_out, ref))      # MIOpen's split-K kernels sum with atomics
        print(f"graph_forward_ms={gfwd:.2f} ({16e3 / gfwd:.0f} img/s)  max_abs_diff_vs_eager={diff:.2e}")
    except Exception as e:  # noqa: BLE001
        print("graph capture of the forward failed:", repr(e)[:300])
    print(f"forward_ms={f:.2f} ({16e3 / f:.0f} img/s)  validation_step_ms={v:.2f}  test_step_ms={te:.2f} ({16e3 / te:.0f} img/s)")


if __name__ == "__main__":
    main()
