"""MIOpen solver choice without a search: the committed user db (`miopen_db/` at the repo root) in IMMEDIATE mode.

This image ships no gfx950 find-db, so `torch.backends.cudnn.benchmark = True` (PL's `Trainer(benchmark=True)`) makes MIOpen measure
every candidate solver of every convolution problem in every process: ~220 s on a fresh machine (38 s with the compiled kernels
cached), and two processes may settle on different solvers where two are within a few per cent -- under data parallelism the
slowest rank sets the step.  `miopen_db/` holds the RESULTS of that search (find-db records, perf-db, compiled-kernel cache;
tools/build_miopen_db.sh, tools/extend_miopen_db.sh) for the convolution problems of the BASELINE configs; in immediate mode
(`benchmark = False`, torch's default) MIOpen ranks solvers by those records: the same solver in every rank and every run, no search,
and the search's speed (profiles/r04_immediate_vs_find.jsonl: YOLOv4 36.47 vs 36.40 ms, RetinaNet 35.54 vs 35.58, YOLOv5-l 35.29 vs
35.19; with an empty db RetinaNet takes 46.8 ms).  A problem that has no record falls back to MIOpen's own ranking."""
import atexit
import os
import shutil
import tempfile

REPO_DB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "miopen_db")


def stage_user_db(src=None):
    """Point MIOpen at a PRIVATE copy of the committed db (one per process: eight ranks never contend for one sqlite file; removed at
    exit).  Call before the first convolution; an already set MIOPEN_USER_DB_PATH wins.  Returns the directory in use or None."""
    if "MIOPEN_USER_DB_PATH" in os.environ:
        return os.environ["MIOPEN_USER_DB_PATH"]
    src = src or REPO_DB
    dst = os.path.join(tempfile.gettempdir(), f"odpl_miopen_db_r{os.environ.get('LOCAL_RANK', '0')}_{os.getpid()}")
    try:
        if os.path.isdir(src):
            shutil.copytree(src, dst, dirs_exist_ok=True)
        os.makedirs(os.path.join(dst, "cache"), exist_ok=True)
        os.environ["MIOPEN_USER_DB_PATH"] = dst
        os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(dst, "cache"))
        atexit.register(shutil.rmtree, dst, ignore_errors=True)
        return dst
    except OSError:
        return None
