#!/usr/bin/env python3
"""Round 4: WHERE does capturing the YOLOv4 backward into a HIP graph die (rc -11 in profiles/r03_probe_step_graph.jsonl)?
Bisection by construction, one child process per case (a capture that crashes must not take the others down), `faulthandler` on in
every child so that the one run that dies leaves its Python frames on stderr, every warm-up on the capture's side stream only (the
round-3 probe ran four eager steps on the DEFAULT stream first, which pins the AccumulateGrad nodes there):
  unit          two Conv2d + BatchNorm + Mish units of the network (odpl_torch_ext BN function), forward + backward
  unit_split    the same with convsplit.enabled (weight gradients on a side stream, an end-of-backward callback)
  plain         the whole YOLOv4, torch / MIOpen ops only (BN epilogues off, head GEMM off, dense criterion backward)
  epilogues     + the BN / activation epilogues of this package
  full          + hosted head convolutions and the structured head backward (the bench's configuration)
  full_r3warm   `full` with round 3's warm-up (four eager steps on the default stream first)
Prints one JSON line per case: rc, last phase reached, replay / eager time, stderr tail."""
import json
import os
import subprocess
import sys
import time

CASES = ["unit", "unit_split", "plain", "epilogues", "full", "full_r3warm"]


def child(case):
    import faulthandler
    faulthandler.enable(all_threads=True)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    bench._stage_miopen_db()
    import torch
    bench.torch = torch
    from objectdetectionpl_amd.LightningFunc.losses import _RegionLossBase
    from objectdetectionpl_amd.LightningFunc.utils import bnact, convsplit
    from objectdetectionpl_amd.LightningFunc.utils.headconv import HeadConv2d
    from objectdetectionpl_amd.model import YOLOV4 as v4
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = False
    torch.manual_seed(0)
    if case in ("plain",):
        v4.ConvUnit.fuse_epilogue = False
    if case in ("plain", "epilogues"):
        HeadConv2d.enabled = False
        _RegionLossBase.fuse_head_backward = False
    convsplit.enabled = case == "unit_split"
    if case.startswith("unit"):
        net = torch.nn.Sequential(v4.ConvUnit(32, 64, 3, 1, "mish"), v4.ConvUnit(64, 64, 1, 1, "mish")).to(dev).train()
        net = net.to(memory_format=torch.channels_last)
        x = torch.randn(4, 32, 52, 52, device=dev).contiguous(memory_format=torch.channels_last)
        params = list(net.parameters())

        def body():
            for p in params:
                p.grad = None
            loss = net(x).square().mean()
            loss.backward()
            return loss.detach()
    else:
        model = v4.YOLOv4([str(i) for i in range(80)], bench.Args()).to(dev).train().to(memory_format=torch.channels_last)
        x, y = bench.synthetic_batch(1, 16, 80, 416, dev)
        x = x.contiguous(memory_format=torch.channels_last)

        def body():
            model.zero_grad(set_to_none=True)
            d = model.criterion(model(x), y)
            d["loss"].backward()
            return d["loss"].detach()

    if case == "full_r3warm":
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
    print(json.dumps({"case": case, "phase": "warm"}), flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = body()
    torch.cuda.synchronize()
    print(json.dumps({"case": case, "phase": "captured"}), flush=True)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 10
    for _ in range(K):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / K
    t0 = time.perf_counter()
    for _ in range(K):
        body()
    torch.cuda.synchronize()
    de = (time.perf_counter() - t0) / K
    print(json.dumps({"case": case, "phase": "done", "replay_ms": round(dt * 1e3, 3), "eager_ms": round(de * 1e3, 3), "out": float(out)}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1])
    else:
        for c in CASES:
            try:
                r = subprocess.run([sys.executable, os.path.abspath(__file__), c], capture_output=True, text=True, timeout=240)
                rc, out, err = r.returncode, r.stdout, r.stderr
            except subprocess.TimeoutExpired as e:
                rc, out, err = "timeout", (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or ""), \
                    (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
            lines = [l for l in out.splitlines() if l.startswith("{")]
            keep = [l for l in err.splitlines() if "UserWarning" not in l and "warn" not in l.lower()]
            print(json.dumps({"case": c, "rc": rc, "last": json.loads(lines[-1]) if lines else None,
                              "stderr_tail": "\n".join(keep)[-1800:] if rc != 0 else ""}), flush=True)
