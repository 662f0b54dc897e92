#!/usr/bin/env python3
"""bench.py -- BASELINE.json metric: images/sec of the YOLOv4 416x416 bs=16/GPU training step.

One "step" = forward (CSPDarknet53/PAN convs on MIOpen) + the HIP region loss (libodpl_hip.so:
assignment, 7 loss terms and d loss/d head) + backward + Adam, on one synthetic COCO-shaped batch
resident in HBM.  N>1: one process per GPU (torchrun), DDP over RCCL/xGMI, weak scaling.

Prints ONE JSON line on rank 0 (contract in the task statement) carrying
  roofline     : the dominant kernel of the loss path, timed with HIP events on its launch stream; `frac` is the LOWER of that and
                 the committed rocprofv3 summary's figure for the same kernel (profiles/)
  loss_path    : the region-loss launches of a step against SURVEY 8d's budget (115.8 MB at 70 % of HBM peak = 20.7 us)
  cpu_baseline : the same training step on the host cores with the CPU oracle as the loss (rank 0, N=1 only)
  secondary    : BASELINE configs 4 (RetinaNet-R50-FPN 800x1333 bs 4) and 5 (YOLOv5-l 640^2 bs 32 bf16) as bounded training-step
                 measurements after the judged workload (rank 0, N=1 only)
  N > 1        : per_rank (step time and host enqueue time of every rank), exposed_comm_ms, allreduce
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# MIOpen ships no gfx950 find-db / kernel-db in this image, so the solver search over the ~300 conv problems of YOLOv4 costs
# ~220 s on a fresh box -- and a search re-MEASURES its candidates in every process, so two ranks (or two runs) may settle on
# different solvers where two are within a few per cent (+-0.7 ms per step, round 3).  The search RESULTS (text find-db + perf-db +
# compiled-kernel cache, produced by tools/build_miopen_db.sh and tools/extend_miopen_db.sh) live under miopen_db/, and the bench runs
# MIOpen's IMMEDIATE mode on them (torch.backends.cudnn.benchmark = False): the solver of every problem is the fastest one RECORDED
# there -- no search at start-up (first step after 2.5 s instead of 38 s), the same choice in every rank and every run, and the
# same speed as the search (36.47 vs 36.40 ms per step on one box, profiles/r04_immediate_vs_find.jsonl).  Each rank works on a
# private copy so that eight processes never contend for one sqlite file (removed again at exit).  `--miopen-find` = the search.
def _stage_miopen_db():
    # (objectdetectionpl_amd/miopen.py without importing the package: the tools call this before they import torch)
    import importlib.util
    spec = importlib.util.spec_from_file_location("_odpl_miopen", os.path.join(REPO, "objectdetectionpl_amd", "miopen.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.stage_user_db()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(n, argv):
    """`python bench.py --gpus N` started by hand (no torchrun): the reference's multi-GPU entry is ONE process start
    (run.py:109-115 with configs/config.yaml:58-61 `gpus: -1`; PL spawns one process per GPU).  This parent never touches
    the GPU (no torch import, no HIP call): it starts one child per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*
    set as torchrun would, relays rank 0's single JSON line, and exits non-zero if any rank does."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    env["MASTER_PORT"] = str(_free_port())
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    env["ODPL_BENCH_CHILD"] = "1"
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        # rank 0's stdout is the JSON channel; the other ranks' stdout goes to our stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    rc = 0
    try:
        while True:                             # a rank that dies takes the job down: its peers would wait in a collective
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes) or any(c not in (None, 0) for c in codes):
                break
            time.sleep(0.1)
    finally:
        for p in procs:                         # exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=20)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    reader.join(timeout=10)
    out0 = b"".join(c for c in chunks if c)
    for r, p in enumerate(procs):
        if p.returncode != 0:
            print(f"[bench launcher] rank {r} exited with code {p.returncode}", file=sys.stderr, flush=True)
            rc = rc or (p.returncode if p.returncode and p.returncode > 0 else 1)
    lines = [l for l in out0.decode(errors="replace").splitlines() if l.strip()]
    if rc == 0 and len(lines) != 1:
        print(f"[bench launcher] rank 0 printed {len(lines)} lines, expected exactly one", file=sys.stderr, flush=True)
        rc = 1
    if rc == 0:
        sys.stdout.write(lines[0] + "\n")
        sys.stdout.flush()
    return rc


HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
STEADY_STATE_CSV = "r04_bench_steady_state.csv" if os.path.exists(os.path.join(REPO, "profiles", "r04_bench_steady_state.csv")) else "r03_bench_steady_state.csv"
LOSS_BUDGET_US = 20.7          # 115 839 360 B at 70 % of 8 TB/s
V4_ANCHORS = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
V4_MASKS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
GPU_BOX_CPU_SHARE = 16          # host threads one GPU's worth of a node may use (the box exposes every core of the host)

torch = None                    # imported by main() in the rank processes only: the launcher parent never loads it
dist = None


class Args:
    model_name = "YOLOv4"
    cls_criterion = "bce_loss"          # configs/config.yaml:30
    coord_criterion = "smooth_l1_loss"  # configs/config.yaml:31-32 (last key wins)
    data_module = "synthetic"
    optimizer = "Adam"
    lr = 1e-3
    weight_decay = 1e-5


def synthetic_batch(seed, B, C, img, device):
    """x ~ U[0,1), 7 targets per image, every index in range."""
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(B, 3, img, img, generator=g)
    n = B * 7
    b = torch.arange(B).repeat_interleave(7).float()
    cls = torch.randint(0, C, (n,), generator=g).float()
    wh = torch.rand(n, 2, generator=g) * 0.48 + 0.02
    xy = torch.rand(n, 2, generator=g) * (1 - wh)
    y = torch.cat([b[:, None], cls[:, None], xy, wh], 1)
    return x.to(device), y.to(device)


def allotted_cores():
    """(threads to use, what bounded them): the affinity mask, a cgroup-v2 CPU quota if one is set, and one GPU's share
    of the node (GPU_BOX_CPU_SHARE) -- the bench runs beside seven other GPUs' worth of work on a full node."""
    n, why = os.cpu_count() or 1, "os.cpu_count"
    try:
        a = len(os.sched_getaffinity(0))
        if a < n:
            n, why = a, "sched_getaffinity"
    except Exception:  # noqa: BLE001
        pass
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()
        if quota != "max":
            q = max(1, int(int(quota) / int(period)))
            if q < n:
                n, why = q, "cgroup cpu.max"
    except Exception:  # noqa: BLE001
        pass
    if n > GPU_BOX_CPU_SHARE:
        n, why = GPU_BOX_CPU_SHARE, f"one GPU's share of the node ({GPU_BOX_CPU_SHARE} of {os.cpu_count()} visible cores)"
    return n, why


def cpu_baseline(B_sample, C, img, steps, warmup=1):
    """The reference's CPU path restated: same network on torch-CPU + the oracle loss + Adam, timed on the host
    (same seeds and shapes as config 2; a BOUNDED sample so the default run stays within minutes)."""
    from oracle import yolo as oracle_yolo
    from objectdetectionpl_amd.model.YOLOV4 import YOLOv4
    cores, why = allotted_cores()
    torch.set_num_threads(cores)
    torch.manual_seed(0)
    model = YOLOv4([str(i) for i in range(C)], Args()).train()
    opt = torch.optim.Adam(model.parameters(), lr=Args.lr, weight_decay=Args.weight_decay)
    x, y = synthetic_batch(1, B_sample, C, img, "cpu")

    def step():
        opt.zero_grad(set_to_none=True)
        out = model(x)
        t0 = time.perf_counter()
        d = oracle_yolo.multiscale_region_loss_v4(out, y, V4_ANCHORS, V4_MASKS, C, img)
        t1 = time.perf_counter()
        d["loss"].backward()
        opt.step()
        return t1 - t0

    for _ in range(warmup):
        step()
    t0 = time.perf_counter()
    loss_s = 0.0
    for _ in range(steps):
        loss_s += step()
    dt = time.perf_counter() - t0
    return {"value": round(B_sample * steps / dt, 3), "unit": "img/s", "cores": cores, "kind": "port",
            "sample": f"{warmup} warm-up + {steps} timed training step(s) of bs={B_sample} (same net on torch-CPU fp32, oracle loss, "
                      f"Adam), {dt:.1f} s on {cores} threads (bounded by {why}); oracle loss forward alone "
                      f"{1e3 * loss_s / steps:.0f} ms/step"}


def rocprof_row(kernel, only_current=False):
    """Average duration of `kernel` in the committed `rocprofv3 --kernel-trace --stats` summary of this command
    (profiles/r04_rocprof_stats_bench.csv; the dense route's expand kernel: the round-2 pass of `--no-fused-head-backward`), so the live
    event timing and the profiler's figure sit side by side."""
    import csv
    files = ("r04_rocprof_stats_bench.csv",) if only_current else ("r04_rocprof_stats_bench.csv", "r02dense_rocprof_stats_bench.csv")
    for name in files:
        try:
            with open(os.path.join(REPO, "profiles", name)) as f:
                for row in csv.DictReader(f):
                    if kernel + "(" in row.get("Name", "") or kernel + "<" in row.get("Name", ""):
                        return {"file": "profiles/" + name, "avg_us": round(float(row["AverageNs"]) / 1e3, 2), "calls": int(row["Calls"])}
        except Exception:  # noqa: BLE001
            continue
    return None


def conv_portion(batch):
    """MFMA utilisation of the convolution portion, reported separately -- from the committed steady-state table of
    this command under rocprofv3 (profiles/<STEADY_STATE_CSV>): MIOpen / CK / Tensile kernel time per step against
    3 x 2 x 30.05 GMAC per image (forward + two backward products, SURVEY a-21) and the 157.3 TFLOP/s fp32 MFMA peak."""
    import csv
    path = os.path.join(REPO, "profiles", STEADY_STATE_CSV)
    try:
        with open(path) as f:
            head = f.readline()
            steps = int(head.split("last ")[1].split(" steps")[0])
            us = sum(float(r["total_us"]) for r in csv.DictReader(f) if r["kernel"].startswith(("igemm", "ck::", "Cijk")))
        ms = us / steps / 1e3
        tf = 3 * 2 * 30.05e9 * batch / (ms * 1e-3) / 1e12
        return {"ms_per_step": round(ms, 3), "tflops": round(tf, 1), "frac_of_fp32_mfma_peak": round(tf / 157.3, 3),
                "source": f"profiles/{STEADY_STATE_CSV} (committed rocprofv3 pass of this command at bs 16; not re-measured in this run)"}
    except Exception:  # noqa: BLE001
        return None


def secondary(steps, log, elapsed_s, budget_s=270.0):
    """BASELINE configs 4 and 5 as bounded training-step measurements (tools/bench_retina_step.py, tools/bench_v5.py: the models,
    criteria and synthetic batches of SURVEY 8d), run in this process AFTER the judged workload's numbers are final, so that the
    driver-timed record carries them too.  MIOpen immediate mode on the committed find-db; each leg inside try / except and skipped
    when the run is already long; nothing here touches `value`."""
    out = {}
    tools = os.path.join(REPO, "tools")
    if tools not in sys.path:
        sys.path.insert(0, tools)
    legs = (("config4", "bench_retina_step", dict(steps=steps, warmup=3)),
            ("config4_small_boxes", "bench_retina_step", dict(steps=steps, warmup=3, gt_per_image=16, wh_range=(0.02, 0.2))),
            ("config5", "bench_v5", dict(steps=steps, warmup=3)))
    t0 = time.perf_counter()
    for name, mod, kw in legs:
        if elapsed_s + (time.perf_counter() - t0) > budget_s:
            out[name] = {"skipped": f"run already at {elapsed_s + time.perf_counter() - t0:.0f} s"}
            continue
        try:
            torch.cuda.empty_cache()
            m = __import__(mod)
            r = m.run(log=log, **kw)
            torch.cuda.synchronize()
            out[name] = r
            log(f"secondary {name}: {r['img_per_s']} img/s ({r['ms_per_step']} ms/step)")
        except Exception as e:  # noqa: BLE001
            out[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            log(f"secondary {name} failed: {out[name]['error']}")
    out["note"] = ("secondary workloads, not the judged metric: config 4 = RetinaNet-R50-FPN 800x1333 bs 4 fp32 (8 boxes / image of 5-50 % of the "
                   "image, SURVEY 8d; config4_small_boxes = 16 boxes of 2-20 %, where the loss-driven step has more pixels to evaluate), "
                   "config 5 = YOLOv5-l 640^2 bs 32, convolutions bf16 -- full training steps (fwd + HIP loss + bwd + fused Adam)")
    return out


def dryrun_rank(json_fd):
    """ODPL_BENCH_DRYRUN=1 (tests/test_bench_launcher.py, no GPU): the rank processes only rendezvous over gloo, sum their
    ranks and rank 0 prints one line -- exercises the launcher / env plumbing / exit-code relay without a device."""
    import torch.distributed as d
    import torch as t
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if os.environ.get("ODPL_BENCH_DRYRUN_FAIL_RANK") == str(rank):
        sys.exit(7)
    d.init_process_group("gloo")
    v = t.tensor([float(1 + rank)])
    d.all_reduce(v)
    d.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps({"dryrun": True, "n_gpus": world, "seed_sum": float(v.item()),
                                       "master": os.environ["MASTER_ADDR"] + ":" + os.environ["MASTER_PORT"]}) + "\n").encode())
    d.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--classes", type=int, default=80)
    ap.add_argument("--img", type=int, default=416)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-batch", type=int, default=16)
    ap.add_argument("--cpu-sample-steps", type=int, default=3)
    ap.add_argument("--bucket-mb", type=int, default=48)
    ap.add_argument("--ddp-impl", default="bucketed", choices=["bucketed", "torch"],
                    help="gradient exchange: own bucketed averager (7 launches/step) or torch DistributedDataParallel")
    ap.add_argument("--overlap-wgrad", action="store_true",
                    help="issue the convolutions' weight-gradient kernels on a side stream (LightningFunc/utils/convsplit.py)")
    ap.add_argument("--force-ddp", action="store_true", help="wrap in DDP (RCCL) even at world size 1: exercises the N>1 code path on one GPU")
    ap.add_argument("--stamp", action="store_true", help="diagnostic: in-kernel 100 MHz stamps of yolo_expand_kernel (first start -> last end)")
    ap.add_argument("--tune", action="append", default=[], metavar="KEY=VALUE", help="libodpl_hip tunable, e.g. yolo_fill_nt=1")
    ap.add_argument("--miopen-find", dest="miopen_find", action="store_true",
                    help="torch.backends.cudnn.benchmark=True: MIOpen re-measures its solvers in every process (~38 s at start-up with the "
                         "committed kernel cache); default: immediate mode on the committed find-db -- same solvers, no search, every rank alike")
    ap.add_argument("--no-miopen-find", dest="miopen_find", action="store_false", help="(the default; kept for older command lines)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the bounded config-4 / config-5 measurements after the judged workload")
    ap.add_argument("--secondary-steps", type=int, default=8)
    ap.add_argument("--comm-probe-steps", type=int, default=4, help="N > 1: steps inside no_sync() that price the exposed gradient exchange")
    ap.add_argument("--no-fused-adam", dest="fused_adam", action="store_false", help="foreach Adam instead of the fused one")
    ap.add_argument("--torch-adam", action="store_true", help="A/B: torch's own fused Adam launches instead of the one-launch HIP step (optim.py)")
    ap.add_argument("--no-channels-last", dest="channels_last", action="store_false", help="NCHW memory format for the network")
    ap.add_argument("--no-head-gemm", dest="head_gemm", action="store_false",
                    help="A/B: head 1x1 convolutions on MIOpen (channels-last heads) instead of utils/headconv.py")
    ap.add_argument("--no-fused-head-backward", dest="fused_head", action="store_false",
                    help="A/B: the criterion hands autograd the dense d loss/d head instead of running the head convolutions' backward itself")
    ap.set_defaults(miopen_find=False)
    a = ap.parse_args()

    # ---- launcher: `python bench.py --gpus N` without torchrun -> this process only starts the N ranks -------------------
    if a.gpus > 1 and "RANK" not in os.environ:
        sys.exit(launch_ranks(a.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if a.gpus > 1 and world != a.gpus:
        sys.exit(f"--gpus {a.gpus} but WORLD_SIZE={world}: start it as `python bench.py --gpus {a.gpus}` or "
                 f"`python -m torch.distributed.run --nproc-per-node {a.gpus} bench.py --gpus {a.gpus} ...`")
    t_start = time.perf_counter()
    # stdout carries exactly ONE line (the JSON): libraries that print banners there (RCCL's version block, gloo, MIOpen) are
    # sent to stderr by pointing fd 1 at fd 2 for the duration of the run; the result is written to the saved descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if os.environ.get("ODPL_BENCH_DRYRUN") == "1":
        return dryrun_rank(json_fd)

    global torch, dist
    _stage_miopen_db()
    import torch as _torch
    import torch.distributed as _dist
    torch, dist = _torch, _dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X (torch.cuda.is_available() is False); there is no CPU fallback")
    # REHEARSAL ONLY: ODPL_BENCH_BACKEND=gloo lets several ranks share one GPU (RCCL refuses duplicate devices) so that the
    # multi-process control flow can be exercised on a 1-GPU box; the judged runs use the default, RCCL ("nccl").
    backend = os.environ.get("ODPL_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(torch.cuda.device_count(), 1)
    elif local_rank >= torch.cuda.device_count():
        sys.exit(f"rank {rank}: LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} GPU(s) visible "
                 f"(RCCL needs one device per rank; ODPL_BENCH_BACKEND=gloo rehearses the control flow on fewer)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_ddp = world > 1 or a.force_ddp
    if use_ddp:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.model.YOLOV4 import YOLOv4
    from objectdetectionpl_amd.parallel import wrap_data_parallel

    if a.overlap_wgrad:
        from objectdetectionpl_amd.LightningFunc.utils import convsplit
        convsplit.enabled = True
    for kv in a.tune:
        k, v = kv.split("=")
        _lib.tune_set(k, int(v))
    torch.backends.cudnn.benchmark = bool(a.miopen_find)
    from objectdetectionpl_amd.LightningFunc.losses import _RegionLossBase
    from objectdetectionpl_amd.LightningFunc.utils.headconv import HeadConv2d
    HeadConv2d.enabled = bool(a.head_gemm)
    _RegionLossBase.fuse_head_backward = bool(a.fused_head)
    torch.manual_seed(0)
    model = YOLOv4([str(i) for i in range(a.classes)], Args()).to(dev).train()
    if a.channels_last:
        model = model.to(memory_format=torch.channels_last)
    crit = model.criterion
    net = wrap_data_parallel(model, dev, bucket_mb=a.bucket_mb, impl=a.ddp_impl) if use_ddp else model
    from objectdetectionpl_amd.optim import Adam as OdplAdam            # what the package's configure_optimizers returns (optimizer.py:6-38)
    OdplAdam.one_launch = not a.torch_adam
    opt = OdplAdam(model.parameters(), lr=Args.lr, weight_decay=Args.weight_decay, fused=bool(a.fused_adam))
    x, y = synthetic_batch(1 + rank, a.batch, a.classes, a.img, dev)       # rank r draws from seed 1 + r
    if a.channels_last:
        x = x.contiguous(memory_format=torch.channels_last)

    def step():
        opt.zero_grad(set_to_none=True)
        out = net(x)
        d = crit(out, y)
        d["loss"].backward()
        opt.step()
        return d

    def fence():
        if use_ddp:
            dist.barrier()
        torch.cuda.synchronize()

    def log(msg):
        if rank == 0:
            print(f"[bench {time.perf_counter() - t_start:7.1f}s] {msg}", file=sys.stderr, flush=True)

    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        log(f"warmup step {i + 1}/{a.warmup} done")     # the first step includes MIOpen solver selection / kernel builds
    fence()
    PROF = ("yolo_assign", "yolo_loss", "yolo_finalize", "yolo_expand", "yolo_patch", "headgrad_stream", "headgrad_finish")
    for k in PROF:
        _lib.profile_enable(k, True)
    stamps = []
    if a.stamp:
        sbuf = torch.zeros(2 * 8192, dtype=torch.int64, device=dev)
        _lib.check(_lib.lib().odpl_debug_set_stamp_buffer(_lib.ptr(sbuf)), "odpl_debug_set_stamp_buffer")
    t0 = time.perf_counter()
    for _ in range(a.steps):
        if a.stamp:
            sbuf.zero_()
        d = step()
        if a.stamp:
            v = sbuf.view(-1, 2)
            v = v[v[:, 0] != 0]
            stamps.append(float(v[:, 1].max() - v[:, 0].min()) / 100.0)      # us: first workgroup start -> last end
    t_enq = time.perf_counter() - t0      # the host has issued every launch of the K steps (the GPU may still be running them)
    fence()
    dt = time.perf_counter() - t0
    prof = {}
    for k in PROF:
        _lib.profile_enable(k, False)
        prof[k] = _lib.profile_collect(k)
    fused = len(prof["headgrad_stream"][0]) > 0        # the criterion ran the head convolutions' backward itself
    roof_key = "headgrad_stream" if fused else "yolo_expand"
    dense_ms, dense_raw_ms = prof[roof_key]
    loss_val = float(d["loss"].item())

    dt_local, per_rank, exposed = dt, None, None
    tt = torch.tensor([dt], dtype=torch.float64, device=dev)
    if use_ddp:
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    dt = float(tt.item())
    if use_ddp:
        # What a poor N-GPU number would be made of (OUTSIDE the timed region, every rank takes part):
        #  per_rank      every rank's own wall time and host enqueue time per step over the K timed steps -- a straggler (another
        #                solver, a throttled GPU, a starved host thread) shows as max >> median; all of them ~equal and high = the exchange
        #  exposed_comm  the same step with the exchange switched off (no_sync: gradients stay local), same process, same buffers:
        #                step - step_without = the part of the gradient exchange that backward did NOT hide (incl. the bucket packs)
        slots = torch.zeros(world, 2, dtype=torch.float64, device=dev)      # every rank fills its own row; one SUM all-reduce gathers them
        slots[rank, 0], slots[rank, 1] = dt_local / a.steps, t_enq / a.steps
        dist.all_reduce(slots)
        allr = [slots[r] for r in range(world)]
        ms = sorted(1e3 * float(v[0]) for v in allr)
        enq = sorted(1e3 * float(v[1]) for v in allr)
        per_rank = {"ms_per_step": {"min": round(ms[0], 3), "median": round(ms[len(ms) // 2], 3), "max": round(ms[-1], 3)},
                    "host_enqueue_ms_per_step": {"min": round(enq[0], 3), "median": round(enq[len(enq) // 2], 3), "max": round(enq[-1], 3)},
                    "by_rank_ms_per_step": [round(1e3 * float(v[0]), 3) for v in allr]}
        if a.comm_probe_steps > 0 and hasattr(net, "no_sync"):
            def timed(k, ctx):
                fence()
                t1 = time.perf_counter()
                with ctx():
                    for _ in range(k):
                        step()
                fence()
                return (time.perf_counter() - t1) / k
            import contextlib
            with_x = timed(a.comm_probe_steps, contextlib.nullcontext)
            without = timed(a.comm_probe_steps, net.no_sync)          # (the replicas drift apart from here on: nothing after this trains)
            ex = torch.tensor([with_x, without], dtype=torch.float64, device=dev)
            dist.all_reduce(ex, op=dist.ReduceOp.MAX)
            exposed = {"ms": round(1e3 * float(ex[0] - ex[1]), 3), "step_ms": round(1e3 * float(ex[0]), 3),
                       "step_without_exchange_ms": round(1e3 * float(ex[1]), 3), "steps": a.comm_probe_steps,
                       "note": "max over ranks of each leg; step - step inside no_sync() = gradient exchange not hidden under backward"}

    allreduce = None
    if use_ddp and world > 1:       # bus bandwidth of the gradient all-reduce, measured OUTSIDE the timed region
        nparam = sum(p.numel() for p in model.parameters())
        flat = torch.zeros(nparam, dtype=torch.float32, device=dev)
        for _ in range(2):
            dist.all_reduce(flat)
        fence()
        t1 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            dist.all_reduce(flat)
        fence()
        ar = torch.tensor([(time.perf_counter() - t1) / reps], dtype=torch.float64, device=dev)
        dist.all_reduce(ar, op=dist.ReduceOp.MAX)
        ar_s = float(ar.item())
        allreduce = {"bytes": 4 * nparam, "ms": round(1e3 * ar_s, 3), "backend": backend,
                     "bus_GBps": round(2 * (world - 1) / world * 4 * nparam / ar_s / 1e9, 1),
                     "note": "one unbucketed fp32 all-reduce of all gradients, not overlapped; in the step it is bucketed and hidden under backward"}
        del flat

    if rank == 0:
        CH = 5 + a.classes
        cells = sum(a.batch * 3 * G * G for G in (a.img // 8, a.img // 16, a.img // 32))
        head_cin = {a.img // 8: 256, a.img // 16: 512, a.img // 32: 1024}          # inputs of conv2 / conv10 / conv18 (model/YOLOV4.py:89-127)
        if fused:
            # headgrad_stream_kernel, one launch for the three heads: read X once (dW of the confidence
            # channels), write dX once, read the compact confidence gradient; fp32
            alg_bytes = sum(2 * 4 * a.batch * G * G * c for G, c in head_cin.items()) + 4 * cells
            roof_kernel = "headgrad_stream_kernel"
        else:
            # yolo_expand_kernel: write every element of the dense d loss/d head once + read the compact confidence gradient
            alg_bytes = 4 * cells * CH + 4 * cells
            roof_kernel = "yolo_expand_kernel"
        traffic, traffic_src = None, None
        try:        # PMC counters cannot be collected from inside this process: committed rocprofv3 --pmc passes
            with open(os.path.join(REPO, "profiles", "pmc_traffic.json")) as f:
                pmc = json.load(f)
            if pmc["workload"] == {"batch": a.batch, "classes": a.classes, "img": a.img}:
                traffic = pmc[roof_kernel]["hbm_bytes"]
                traffic_src = "committed PMC pass (" + pmc.get("source", "profiles/pmc_traffic.json") + "), not measured in this run"
        except Exception:  # noqa: BLE001
            pass
        avg_ms = sum(dense_ms) / max(len(dense_ms), 1)
        live = alg_bytes / (avg_ms * 1e-3) / 1e9 if dense_ms else None
        rp = rocprof_row(roof_kernel)
        rp_gbs = None if rp is None else alg_bytes / (rp["avg_us"] * 1e-6) / 1e9
        # `achieved` / `frac`: the LOWER of the two clocks.  The live figure is (stop - start) - (start - pre) of HIP events on the launch
        # stream; rocprofv3's duration for the same kernel runs ~2-3 us longer (its begin stamp precedes the first wave).  The contract's
        # field carries the pessimistic one, so that it follows from the committed summary; the other stays beside it.
        cands = [(v, src) for v, src in ((live, "live HIP events on the launch stream, event-pair cost subtracted"),
                                         (rp_gbs, None if rp is None else "rocprofv3 --kernel-trace --stats average (" + rp["file"] + ")")) if v]
        achieved, frac_src = min(cands) if cands else (None, None)
        launches_us = {k: round(1e3 * sum(v[0]) / len(v[0]), 2) for k, v in prof.items() if v[0]}
        loss_keys = [k for k in launches_us if k.startswith("yolo_") and k not in ("yolo_expand", "yolo_patch")]
        # (the profile key "yolo_loss" is the fused assignment + loss launch up to 512 target rows: yolo_fused_kernel)
        rp_loss = {k: (rocprof_row("yolo_fused_kernel", True) if k == "yolo_loss" and "yolo_assign" not in launches_us else None)
                   or rocprof_row(k + "_kernel", True) for k in loss_keys}
        loss_live = round(sum(launches_us[k] for k in loss_keys), 2)
        loss_rp = round(sum(v["avg_us"] for v in rp_loss.values()), 2) if loss_keys and all(v is not None for v in rp_loss.values()) else None
        loss_path = {"launches": loss_keys, "live_us": {k: launches_us[k] for k in loss_keys}, "live_sum_us": loss_live,
                     "rocprof_us": {k: (None if v is None else v["avg_us"]) for k, v in rp_loss.items()}, "rocprof_sum_us": loss_rp,
                     "budget_us": LOSS_BUDGET_US, "within_budget": None if loss_rp is None else bool(loss_rp <= LOSS_BUDGET_US),
                     "survey_8d_bytes": 2 * 4 * cells * CH,
                     "frac_of_hbm_peak_on_survey_bytes": None if not (loss_rp or loss_live) else
                     round(2 * 4 * cells * CH / ((loss_rp or loss_live) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                     "note": "assignment + loss (+ finalize) of one step; SURVEY 8d prices the reference's dense read + gradient write "
                             "(115.8 MB) at 70 % of 8 TB/s = 20.7 us; the launches touch 2.7 MB and are latency-bound"}
        res = {
            "metric": "images/sec training step (YOLOv4 416x416, bs=16/GPU)",
            "value": round(a.batch * world * a.steps / dt, 2), "unit": "img/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(1e3 * dt / a.steps, 3),
            "host_enqueue_ms_per_step": round(1e3 * t_enq / a.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"YOLOv4 COCO-synthetic {a.img}x{a.img} bs={a.batch}/GPU training step "
                                   f"(fwd + HIP region loss + bwd + Adam), {a.classes} classes, 7 targets/img, rank r seeded 1+r",
                       "global_batch": a.batch * world, "parallelism": f"dp{world}", "loss_last": round(loss_val, 4)},
            "roofline": {"kernel": roof_kernel, "bound": "hbm",
                         "achieved": None if achieved is None else round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 4), "frac_source": frac_src,
                         "frac_live_events": None if live is None else round(live / HBM_PEAK_GBS, 4),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_kernel_us": round(avg_ms * 1e3, 2),
                         "avg_event_pair_us_uncorrected": round(1e3 * sum(dense_raw_ms) / max(len(dense_raw_ms), 1), 2),
                         "rocprof_avg_kernel_us": None if rp is None else rp["avg_us"],
                         "rocprof_frac": None if rp_gbs is None else round(rp_gbs / HBM_PEAK_GBS, 4),
                         "rocprof_source": None if rp is None else rp["file"],
                         "launches": len(dense_ms),
                         "loss_launches_us": launches_us,
                         "loss_launches_sum_us": round(sum(v for k, v in launches_us.items() if k.startswith("yolo_")), 2),
                         "head_backward_launches_sum_us": round(sum(v for k, v in launches_us.items() if k.startswith("headgrad_")), 2),
                         "replaces": ("MIOpen backward of the three head convolutions + the dense 57.9 MB d loss/d head stream "
                                      "(0.89 ms + 11 us, tools/probe_headconv.py)") if fused else None,
                         **({"in_kernel_stamp_us": round(sorted(stamps)[len(stamps) // 2], 2)} if stamps else {}),
                         "algorithmic_bytes": alg_bytes, "survey_8d_bytes": 2 * 4 * cells * CH},
        }
        res["loss_path"] = loss_path
        res["config"]["miopen"] = "solver search per process (cudnn.benchmark)" if a.miopen_find else "immediate mode on the committed find-db (miopen_db/)"
        if allreduce is not None:
            res["allreduce"] = allreduce
        if per_rank is not None:
            res["per_rank"] = per_rank
        if exposed is not None:
            res["exposed_comm_ms"] = exposed["ms"]
            res["exposed_comm"] = exposed
        if a.batch == 16 and a.img == 416:
            cp = conv_portion(a.batch)
            if cp is not None:
                res["conv_portion"] = cp
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(a.cpu_sample_batch, a.classes, a.img, a.cpu_sample_steps)
        if world == 1 and not use_ddp and not a.no_secondary and a.batch == 16 and a.img == 416:
            del d
            model.zero_grad(set_to_none=True)
            opt.state.clear()
            res["secondary"] = secondary(a.secondary_steps, log, time.perf_counter() - t_start)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(res) + "\n").encode())
    if use_ddp:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
