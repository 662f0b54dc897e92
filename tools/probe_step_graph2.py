#!/usr/bin/env python3
"""Which part of the YOLOv4 training step can be captured into a HIP graph on this stack?  Each stage runs in its own process (a
capture that crashes must not take the others down): fwd = the network's forward only; fwdloss = + criterion; bwd = + backward;
step = + fused Adam (capturable).  Prints one JSON line per stage: ok / error text / replay time."""
import json
import os
import subprocess
import sys
import time

STAGES = ["fwd", "fwdloss", "bwd", "step"]


def child(stage):
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench._stage_miopen_db()
    import torch
    bench.torch = torch
    from objectdetectionpl_amd.model.YOLOV4 import YOLOv4
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model = YOLOv4([str(i) for i in range(80)], bench.Args()).to(dev).train().to(memory_format=torch.channels_last)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, weight_decay=1e-5, fused=True, capturable=True)
    x, y = bench.synthetic_batch(1, 16, 80, 416, dev)
    x = x.contiguous(memory_format=torch.channels_last)

    def body():
        if stage == "fwd":
            with torch.no_grad():
                return model(x)[0].sum()
        if stage == "fwdloss":
            with torch.no_grad():
                return model.criterion(model(x), y)["loss"]
        opt.zero_grad(set_to_none=True)
        d = model.criterion(model(x), y)
        d["loss"].backward()
        if stage == "step":
            opt.step()
        return d["loss"].detach()

    for _ in range(4):
        body()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            body()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    print(json.dumps({"stage": stage, "phase": "warm"}), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = body()
    torch.cuda.synchronize()
    print(json.dumps({"stage": stage, "phase": "captured"}), flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 20
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        body()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / K
    print(json.dumps({"stage": stage, "phase": "done", "replay_ms": round(dt * 1e3, 3), "eager_ms": round(de * 1e3, 3), "out": float(out)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for s in STAGES:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), s], capture_output=True, text=True, timeout=500)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            print(json.dumps({"stage": s, "rc": r.returncode, "last": lines[-1] if lines else None, "stderr_tail": r.stderr[-400:] if r.returncode else ""}), flush=True)
