"""Data-parallel host for the hot path: one process per GPU, gradients averaged with RCCL over xGMI.

The reference has no explicit collective: `Trainer(gpus=-1)` (run.py:109, configs/config.yaml:58-61)
makes PL 1.2.6 wrap the module in DistributedDataParallel.  This is that exchange, re-hosted:
  * the per-image detector kernels need no collective (targets carry a LOCAL image index, loss means
    are per rank, no SyncBN) -- the only exchange is the mean of fp32 gradients, 257.5 MB for YOLOv4;
  * `GradAverager` (default): parameters are grouped into flat fp32 buckets in reverse registration order
    (~ the order backward produces them).  When the last gradient of a bucket lands, ONE multi-tensor copy
    packs the bucket, `.grad` of its parameters is re-pointed at views of the flat buffer, and an asynchronous
    all-reduce(AVG) of the whole bucket is issued on RCCL's stream while backward keeps running; a callback at
    the end of backward waits for the outstanding collectives.  torch's DDP does the same exchange but, with
    `zero_grad(set_to_none=True)`, pays one copy + one scale kernel PER PARAMETER per step (609 launches, 2 ms
    for YOLOv4, `profiles/r01_ddp_*`); here it is 7 launches;
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


class _Bucket:
    __slots__ = ("params", "flat", "views", "pending", "work")

    def __init__(self, params):
        self.params = params
        n = sum(p.numel() for p in params)
        self.flat = torch.zeros(n, dtype=params[0].dtype, device=params[0].device)
        self.views, o = [], 0
        for p in params:
            # same sizes AND strides as the parameter (conv weights may be channels-last): fused optimisers require
            # grad and param layouts to match, and the pack becomes a straight copy
            self.views.append(self.flat[o:o + p.numel()].as_strided(p.size(), p.stride()))
            o += p.numel()
        self.pending = len(params)
        self.work = None


class GradAverager:
    """Bucketed, overlapped mean of gradients over the process group (see the module docstring)."""

    def __init__(self, params, bucket_mb=48, first_bucket_mb=4, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.avg_op = dist.ReduceOp.AVG if dist.get_backend(group) == "nccl" else None     # gloo: SUM, then one scale per bucket
        params = [p for p in params if p.requires_grad][::-1]
        self.buckets, cur, cur_bytes, cap = [], [], 0, first_bucket_mb * 2 ** 20
        for p in params:
            if cur and (cur_bytes + p.numel() * p.element_size() > cap or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(_Bucket(cur))
                cur, cur_bytes, cap = [], 0, bucket_mb * 2 ** 20
            cur.append(p)
            cur_bytes += p.numel() * p.element_size()
        if cur:
            self.buckets.append(_Bucket(cur))
        self._bucket_of = {}
        self.enabled = True
        self._armed = False
        self._handles = []
        for b in self.buckets:
            for p in b.params:
                self._bucket_of[p] = b
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def no_sync(self):
        """Context manager for gradient accumulation: inside it gradients accumulate locally and nothing is exchanged; the
        first backward outside it averages the accumulated gradients (the `DistributedDataParallel.no_sync` contract)."""
        import contextlib

        @contextlib.contextmanager
        def _cm():
            prev, self.enabled = self.enabled, False
            try:
                yield
            finally:
                self.enabled = prev
        return _cm()

    def _on_grad(self, p):
        if not self.enabled:
            return
        if not self._armed:                       # first gradient of this backward pass: arrange the final wait
            self._armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.finish)
        b = self._bucket_of[p]
        b.pending -= 1
        if b.pending == 0:
            self._launch(b)

    def _launch(self, b):
        if b.flat.is_cuda:
            from .LightningFunc.utils import convsplit
            convsplit.join(b.flat.device)                               # weight gradients issued on the side stream (if any)
        torch._foreach_copy_(b.views, [p.grad for p in b.params])      # one multi-tensor kernel per bucket
        for p, v in zip(b.params, b.views):
            p.grad = v                                                  # the optimiser reads the averaged flat buffer
        if self.avg_op is not None:
            b.work = dist.all_reduce(b.flat, op=self.avg_op, group=self.group, async_op=True)
        else:
            b.work = dist.all_reduce(b.flat, group=self.group, async_op=True)

    def finish(self):
        """End of backward: wait for the outstanding collectives (the current stream waits; the host does not block on RCCL).
        A bucket that did not fill (a parameter without a gradient this step) is an error -- but the exchange state is reset
        first, so the next backward starts clean instead of optimising on half-reduced buffers."""
        unfilled = [i for i, b in enumerate(self.buckets) if b.pending != 0]
        try:
            for b in self.buckets:
                if b.work is not None:
                    b.work.wait()
                    if self.avg_op is None:
                        b.flat.div_(self.world)
        finally:
            for b in self.buckets:
                b.work, b.pending = None, len(b.params)
            self._armed = False
        if unfilled:
            raise RuntimeError(f"gradient bucket(s) {unfilled} did not fill: every parameter must receive a gradient each step "
                               "(static graph, like the reference's training step)")

    def remove(self):
        for h in self._handles:
            h.remove()
        self._handles = []


class DataParallelHost(torch.nn.Module):
    """`.module` + forward passthrough (the DDP surface the training loops use); gradients are averaged by `GradAverager`.
    broadcast_buffers (DDP's default, hence PL's): before every forward that will synchronise gradients the module's
    buffers (BatchNorm running statistics, num_batches_tracked) are overwritten with rank 0's, one coalesced broadcast per
    dtype -- so per-rank validation sees rank 0's statistics, as with the reference under PL's DDP."""

    def __init__(self, module, bucket_mb=48, first_bucket_mb=4, group=None, broadcast_buffers=True):
        super().__init__()
        self.module = module
        self.group = group
        self.broadcast_buffers = broadcast_buffers
        self._groups = None
        with torch.no_grad():                      # replicas start from rank 0's weights and buffers, like DDP
            for t in list(module.parameters()) + list(module.buffers()):
                dist.broadcast(t, 0, group=group)
        self.averager = GradAverager(module.parameters(), bucket_mb, first_bucket_mb, group)

    def _buffer_groups(self):
        """buffers grouped by (dtype, device), with their sizes -- built once (the module's buffer set is static)"""
        if self._groups is None:
            by = {}
            for b in self.module.buffers():
                by.setdefault((b.dtype, b.device), []).append(b)
            self._groups = [(bufs, [b.numel() for b in bufs]) for bufs in by.values()]
        return self._groups

    def sync_buffers(self):
        if not self.broadcast_buffers or not self.averager.enabled:          # inside no_sync DDP skips it too
            return
        rank0 = dist.get_rank(self.group) == 0
        with torch.no_grad():
            for bufs, sizes in self._buffer_groups():
                flat = torch.cat([b.reshape(-1) for b in bufs])               # one pack, one broadcast, one unpack per dtype
                dist.broadcast(flat, 0, group=self.group)
                if not rank0:
                    torch._foreach_copy_(bufs, [v.view_as(b) for v, b in zip(flat.split(sizes), bufs)])

    def forward(self, *a, **k):
        self.sync_buffers()
        return self.module(*a, **k)

    def no_sync(self):
        return self.averager.no_sync()


def wrap_data_parallel(model, device=None, bucket_mb=48, first_bucket_mb=4, broadcast_buffers=True, impl="bucketed"):
    """Data-parallel wrapper; BN statistics stay per rank (no SyncBN), as in the reference.
    impl="bucketed": `DataParallelHost` (default).  impl="torch": torch's DistributedDataParallel with bucket views."""
    if impl == "bucketed":
        return DataParallelHost(model, bucket_mb, first_bucket_mb, broadcast_buffers=broadcast_buffers)
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
