"""bench.py started by hand at --gpus N (no torchrun) must start the N ranks itself BEFORE any GPU call, relay rank 0's
single JSON line and exit non-zero when any rank fails (the reference's multi-GPU entry is one
process start: run.py:109-115 + configs/config.yaml:58-61).  ODPL_BENCH_DRYRUN=1 makes the ranks rendezvous over gloo
without touching a device, so the launcher itself is covered here on the CPU; tests/test_gpu_bench_launcher.py runs the
real step with two ranks on a GPU box."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(n, extra_env):
    env = dict(os.environ, ODPL_BENCH_DRYRUN="1", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", str(n)], env=env,
                          capture_output=True, text=True, timeout=300)


def test_launcher_starts_ranks_and_relays_one_line():
    r = _run(3, {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 3 and d["seed_sum"] == 1 + 2 + 3           # ranks 0..2 each contributed 1 + rank
    assert d["master"].startswith("127.0.0.1:")


def test_launcher_propagates_a_failing_rank():
    r = _run(2, {"ODPL_BENCH_DRYRUN_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert r.stdout.strip() == ""                                     # no JSON line from a failed job
    assert "rank 1 exited with code 7" in r.stderr


def test_parent_never_imports_torch():
    """The launcher parent must not initialise the GPU: it does not even import torch."""
    code = ("import sys, runpy; sys.argv=['bench.py','--gpus','2']; "
            "import bench; bench.launch_ranks=lambda n, argv: (print('torch' in sys.modules), 0)[1]; bench.main()")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-c", code], cwd=REPO, env=env, capture_output=True, text=True, timeout=120)
    assert r.stdout.strip().splitlines()[-1] == "False", (r.stdout, r.stderr[-1000:])
