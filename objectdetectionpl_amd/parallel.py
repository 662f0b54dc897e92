"""Data-parallel host for the hot path: one process per GPU, gradients averaged with RCCL over xGMI.

The reference has no explicit collective: `Trainer(gpus=-1)` (run.py:109, configs/config.yaml:58-61)
makes PL 1.2.6 wrap the module in DistributedDataParallel.  This is that exchange, re-hosted:
  * the per-image detector kernels need no collective (targets carry a LOCAL image index, loss means
    are per rank, no SyncBN) -- the only exchange is the mean of fp32 gradients, 257.5 MB for YOLOv4;
  * buckets are views of the gradients (`gradient_as_bucket_view`), filled in reverse parameter order
    and all-reduced from autograd hooks while the rest of backward still runs;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): a ring is per-link bound, so buckets are
    kept large (default 48 MB -> 6 buckets for YOLOv4) to amortise per-collective latency, and the
    first (head/neck) bucket is small so the exchange starts early.
backend "nccl" IS RCCL on ROCm; "gloo" is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist
from torch.nn.parallel import DistributedDataParallel


def init_process_group_from_env(device=None):
    """RANK / WORLD_SIZE / MASTER_* from the environment (torchrun); returns (rank, world)."""
    if not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = "nccl" if (device is not None and torch.device(device).type == "cuda") else "gloo"
        kw = {"device_id": torch.device(device)} if backend == "nccl" else {}
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def wrap_data_parallel(model, device=None, bucket_mb=48, first_bucket_mb=4, broadcast_buffers=True):
    """DDP with bucket views and overlap; BN statistics stay per rank (no SyncBN), as in the reference."""
    dev = None if device is None else torch.device(device)
    ids = [dev.index] if dev is not None and dev.type == "cuda" else None
    # the first bucket's size is a torch-internal default (1 MiB); raise it so the tail of backward is
    # not a train of tiny collectives
    try:
        import torch.distributed as _d
        _d._DEFAULT_FIRST_BUCKET_BYTES = int(first_bucket_mb * 1024 * 1024)
    except Exception:  # noqa: BLE001
        pass
    return DistributedDataParallel(model, device_ids=ids, output_device=None if ids is None else ids[0],
                                   bucket_cap_mb=bucket_mb, gradient_as_bucket_view=True,
                                   broadcast_buffers=broadcast_buffers, static_graph=True)


def shard_batch(x, y, rank, world):
    """Split a global (x[B,...], y[nT,6]) batch by image for `rank`, re-basing the local image index
    (dataset collate format, dataset/Pascal.py:176-186).  B must divide by world."""
    B = x.shape[0]
    if B % world:
        raise ValueError(f"global batch {B} not divisible by world size {world}")
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    keep = (y[:, 0] >= lo) & (y[:, 0] < hi)
    ys = y[keep].clone()
    ys[:, 0] -= lo
    return x[lo:hi], ys
