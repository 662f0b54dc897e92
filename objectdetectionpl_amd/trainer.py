"""The subset of `pytorch_lightning.Trainer` that `run.py:109-118` drives -- `Trainer.from_argparse_args(config, callbacks=[...],
logger=...)`, `.tune`, `.fit(model, datamodule=dm)`, `.test(model, datamodule=dm)` -- for machines without PL (it is not
installable here).  `objectdetectionpl_amd/compat/pytorch_lightning` re-exports it under PL's names.

One process per GPU.  Multi-GPU is ONE process start, as in the reference (`configs/config.yaml:58-61` `gpus: -1`): when more
than one device is requested and no launcher set RANK, `fit` re-runs the script once per extra device (PL's own `ddp` plugin
does the same) and becomes rank 0.  Gradients are averaged by `parallel.wrap_data_parallel` (bucketed all-reduce over RCCL,
overlapped with backward); buffers are re-broadcast from rank 0 before every synchronising forward (DDP's
`broadcast_buffers=True`, PL's default); micro-steps that do not end in an optimiser step run under `no_sync`;
`val_loss` is averaged over the ranks before the scheduler, checkpoint and early-stopping decisions, so every rank takes the
same ones.  Not re-implemented: PL's profiler, precision plugins, multi-optimizer loops."""
import contextlib
import inspect
import os
import subprocess
import sys

import torch

from . import parallel


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _ForwardVia:
    """`self` for the model's step functions with `forward` routed through the data-parallel wrapper; everything else is the
    model itself (attributes set by the step functions land on the model)."""

    def __init__(self, model, net):
        object.__setattr__(self, "_m", model)
        object.__setattr__(self, "_n", net)

    def forward(self, *a, **k):
        return self._n(*a, **k)

    __call__ = forward

    def __getattr__(self, k):
        return getattr(self._m, k)

    def __setattr__(self, k, v):
        setattr(self._m, k, v)


class Trainer:
    def __init__(self, max_epochs=1000, min_epochs=1, limit_train_batches=1.0, limit_val_batches=1.0, limit_test_batches=1.0,
                 accumulate_grad_batches=1, callbacks=None, logger=None, gpus=None, auto_select_gpus=False,
                 num_sanity_val_steps=2, auto_lr_find=False, auto_scale_batch_size=False, device=None, accelerator=None,
                 num_processes=1, ddp_impl="bucketed", checkpoint_callback=True, **_ignored):
        self.max_epochs, self.min_epochs = int(max_epochs), int(min_epochs)
        self.limits = dict(train=limit_train_batches, val=limit_val_batches, test=limit_test_batches)
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.callbacks = list(callbacks or [])
        self.logger = logger if logger not in (True, False) else None
        self.num_sanity_val_steps = int(num_sanity_val_steps or 0)
        self.auto_lr_find, self.auto_scale_batch_size = auto_lr_find, auto_scale_batch_size
        self.ddp_impl = ddp_impl
        self.cpu = accelerator == "ddp_cpu" or (device is not None and torch.device(device).type == "cpu")
        # how many ranks were asked for (the launcher's WORLD_SIZE wins)
        if self.cpu:
            want = int(num_processes or 1)
        elif gpus in (None, 0, "0"):
            want = 1
        elif gpus in (-1, "-1"):
            want = max(torch.cuda.device_count(), 1)          # counting devices does not initialise the GPU
        elif isinstance(gpus, (list, tuple)):
            want = len(gpus)
        else:
            want = int(gpus)
        self.requested_world = want
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        self.device = torch.device("cpu") if self.cpu else torch.device(device or ("cuda:%d" % self.local_rank))
        self.checkpoint_callback = next((c for c in self.callbacks if hasattr(c, "best_model_path")), None)
        self.callback_metrics, self.optimizers, self.lr_schedulers = {}, [], []
        self.current_epoch, self.global_step, self.should_stop, self.running_sanity_check = 0, 0, False, False
        self._children = []

    # -- construction from the flattened yaml Namespace (run.py:26-39, 109) ---------------------------------------------
    @classmethod
    def from_argparse_args(cls, args, **kw):
        names = [p for p in inspect.signature(cls.__init__).parameters if p not in ("self", "_ignored")]
        have = vars(args) if not isinstance(args, dict) else args
        merged = {k: have[k] for k in names if k in have}
        merged.update(kw)
        return cls(**merged)

    @property
    def is_global_zero(self):
        return self.rank == 0

    # -- one process start for N devices ----------------------------------------------------------------------------------
    def _launch_ranks(self):
        """PL's `ddp` plugin: re-run this very script once per extra device with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set;
        this process continues as rank 0.  The children are ordinary child processes (never an exec of this one)."""
        n = self.requested_world
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        os.environ["MASTER_ADDR"] = "127.0.0.1"
        os.environ["MASTER_PORT"] = str(_free_port())
        os.environ["WORLD_SIZE"] = os.environ["LOCAL_WORLD_SIZE"] = str(n)
        for r in range(1, n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), ODPL_TRAINER_CHILD="1")
            self._children.append(subprocess.Popen([sys.executable] + sys.argv, env=env))
        os.environ["RANK"] = os.environ["LOCAL_RANK"] = "0"
        self.world, self.rank, self.local_rank = n, 0, 0
        import atexit
        atexit.register(self._reap)

    def _reap(self):
        for p in self._children:
            try:
                p.wait(timeout=600)
            except subprocess.TimeoutExpired:
                p.kill()
        self._children = []

    # -- helpers ------------------------------------------------------------------------------------------------------------
    def _limit(self, loader, kind):
        lim = self.limits[kind]
        if lim is None:
            return None
        if isinstance(lim, float):
            if lim >= 1.0:
                return None
            try:
                return max(1, int(len(loader) * lim))
            except TypeError:
                return None
        return int(lim)

    def _take(self, loader, kind):
        """-> (index, batch on the device, is_last) with one batch of look-ahead so that the last batch is known."""
        lim = self._limit(loader, kind)
        prev = None
        n = 0
        for batch in loader:
            if lim is not None and n >= lim:
                break
            if prev is not None:
                yield n - 1, prev, False
            prev = tuple(t.to(self.device, non_blocking=True) if torch.is_tensor(t) else t for t in batch)
            n += 1
        if prev is not None:
            yield n - 1, prev, True

    def _setup_optimizers(self, model):
        """`configure_optimizers()` in the forms PL accepts: optimizer | (optimizers, schedulers) | dict(optimizer, lr_scheduler,
        monitor) (the reference returns the dict: optimizer.py:34-38).  Without the hook: fused Adam on args.lr."""
        self.optimizers, self.lr_schedulers = [], []
        a = getattr(model, "args", None)
        cfg = None
        if hasattr(model, "configure_optimizers") and (a is None or (hasattr(a, "optimizer") and hasattr(a, "lr_scheduler"))):
            cfg = model.configure_optimizers()
        if cfg is None:
            kw = {"fused": True} if self.device.type == "cuda" else {}
            cfg = torch.optim.Adam(model.parameters(), lr=getattr(a, "lr", 1e-3), weight_decay=getattr(a, "weight_decay", 0.0), **kw)
        if isinstance(cfg, dict):
            self.optimizers = [cfg["optimizer"]]
            sch = cfg.get("lr_scheduler")
            if sch is not None:
                d = dict(sch) if isinstance(sch, dict) else {"scheduler": sch}
                d.setdefault("monitor", cfg.get("monitor", "val_loss"))
                d.setdefault("interval", "epoch")
                self.lr_schedulers = [d]
        elif isinstance(cfg, (tuple, list)) and len(cfg) == 2 and isinstance(cfg[0], (tuple, list)):
            self.optimizers = list(cfg[0])
            self.lr_schedulers = [{"scheduler": s, "monitor": "val_loss", "interval": "epoch"} for s in cfg[1]]
        elif isinstance(cfg, (tuple, list)):
            self.optimizers = list(cfg)
        else:
            self.optimizers = [cfg]
        if len(self.optimizers) != 1:
            raise NotImplementedError("one optimizer (the reference's configure_optimizers, optimizer.py:6-38)")
        return self.optimizers[0]

    def _mean_over_ranks(self, t):
        """mean of a scalar over the data-parallel ranks (every rank then decides on the same number)."""
        if self.world > 1 and torch.distributed.is_initialized():
            t = t.detach().clone().float()
            torch.distributed.all_reduce(t)
            t /= self.world
        return t

    def _call(self, hook, *a):
        for c in self.callbacks:
            fn = getattr(c, hook, None)
            if fn is not None:
                fn(self, *a)

    def _attach(self, model):
        model.trainer = self
        if self.logger is not None and getattr(model, "logger", None) is not self.logger:
            try:
                model.logger = self.logger
            except AttributeError:
                pass                                      # a LightningModule's `logger` property already reads trainer.logger
        if self.checkpoint_callback is not None and getattr(model, "checkpoint_callback", None) is None:
            model.checkpoint_callback = self.checkpoint_callback

    def _set_progress(self, model):
        for k, v in (("current_epoch", self.current_epoch), ("global_step", self.global_step)):
            try:
                setattr(model, k, v)
            except AttributeError:
                pass

    def _validate(self, model, vstep, datamodule, max_batches=None):
        model.eval()
        vouts = []
        for i, b, _last in self._take(datamodule.val_dataloader(), "val"):
            if max_batches is not None and i >= max_batches:
                break
            vouts.append(vstep(b, i))
        return vouts

    # -- fit ----------------------------------------------------------------------------------------------------------------
    def fit(self, model, datamodule=None, train_dataloader=None, val_dataloaders=None):
        if datamodule is None:
            datamodule = train_dataloader
        if self.requested_world > 1 and "RANK" not in os.environ:
            self._launch_ranks()
        if self.world > 1:
            parallel.init_process_group_from_env(self.device)
        if hasattr(datamodule, "setup"):
            datamodule.setup("fit")
        model.to(self.device).train()
        net = parallel.wrap_data_parallel(model, self.device, impl=self.ddp_impl) if self.world > 1 else model
        if net is not model:                                              # forward goes through the wrapper (gradient hooks, buffer broadcast)
            via = _ForwardVia(model, net)
            tstep = lambda b, i: type(model).training_step(via, b, i)      # noqa: E731
            vstep = lambda b, i: type(model).validation_step(via, b, i)    # noqa: E731
        else:
            tstep, vstep = model.training_step, model.validation_step
        opt = self._setup_optimizers(model)
        self._attach(model)
        history = []
        self.should_stop, self.global_step = False, 0
        self._call("on_fit_start", model)
        if self.num_sanity_val_steps > 0:
            self.running_sanity_check = True
            with torch.no_grad():
                self._validate(model, vstep, datamodule, self.num_sanity_val_steps)
            self.running_sanity_check = False
        for epoch in range(self.max_epochs):
            self.current_epoch = epoch
            self._set_progress(model)
            model.train()
            self._call("on_train_epoch_start", model)
            outs = []
            pending = 0                                                   # micro-batches since the last optimiser step
            for i, batch, is_last in self._take(datamodule.train_dataloader(), "train"):
                steps_now = (i + 1) % self.accumulate == 0 or is_last      # PL also steps on the last batch of the epoch
                sync = contextlib.nullcontext() if steps_now or not hasattr(net, "no_sync") else net.no_sync()
                with sync:
                    out = tstep(batch, i)
                    (out["loss"] / self.accumulate).backward()
                pending += 1
                if steps_now:
                    opt.step()
                    opt.zero_grad(set_to_none=True)
                    pending = 0
                self.global_step += 1
                self._set_progress(model)
                outs.append({k: v.detach() for k, v in out.items() if torch.is_tensor(v)})
                self._call("on_train_batch_end", model, out, batch, i)
            if outs and hasattr(model, "training_epoch_end"):
                history.append(("train", epoch, model.training_epoch_end(outs)))
            self._call("on_train_epoch_end", model, outs)
            with torch.no_grad():
                vouts = self._validate(model, vstep, datamodule)
            if vouts:
                val_loss = self._mean_over_ranks(torch.stack([o["loss"].detach().float() for o in vouts]).mean())
                self.callback_metrics["val_loss"] = val_loss
                for k, v in getattr(model, "_logged_metrics", {}).items():
                    if k != "val_loss":
                        self.callback_metrics[k] = v
                if self.logger is not None:
                    self.logger.log_metrics({"val_loss": val_loss}, step=self.global_step)
            for d in self.lr_schedulers:                                   # epoch-level schedulers; the plateau one on its monitor
                sch = d["scheduler"]
                if isinstance(sch, torch.optim.lr_scheduler.ReduceLROnPlateau):
                    if d["monitor"] in self.callback_metrics:
                        sch.step(float(self.callback_metrics[d["monitor"]]))
                else:
                    sch.step()
            if vouts:
                self._call("on_validation_end", model)                     # ModelCheckpoint / EarlyStopping read callback_metrics
                if hasattr(model, "validation_epoch_end"):
                    history.append(("val", epoch, model.validation_epoch_end(vouts)))
            if self.should_stop and epoch + 1 >= self.min_epochs:
                break
        self._call("on_fit_end", model)
        if self.logger is not None:
            self.logger.save()
        if net is not model and hasattr(net, "averager"):
            net.averager.remove()
        return history

    # -- test ---------------------------------------------------------------------------------------------------------------
    def test(self, model=None, datamodule=None, test_dataloaders=None):
        if datamodule is None:
            datamodule = test_dataloaders
        if hasattr(datamodule, "setup"):
            datamodule.setup("test")
        self._attach(model)
        model.to(self.device).eval()
        outs = [model.test_step(b, i) for i, b, _ in self._take(datamodule.test_dataloader(), "test")]
        res = model.test_epoch_end(outs) if outs else {}
        self._call("on_test_end", model)
        if self.logger is not None:
            self.logger.save()
        return res

    # -- tune (run.py:112-113: only when `tune: True`) --------------------------------------------------------------------------
    def tune(self, model, datamodule=None, **_):
        """`auto_scale_batch_size` ('power': double `datamodule.batch_size` while a step fits) and `auto_lr_find` (an
        exponential learning-rate sweep; the rate with the steepest loss decrease is written to `model.args.lr`).  The model's
        weights are restored afterwards."""
        res = {}
        if self.auto_scale_batch_size:
            res["batch_size"] = self._scale_batch_size(model, datamodule)
        if self.auto_lr_find:
            res["lr"] = self._lr_find(model, datamodule)
        return res

    def _probe_steps(self, model, datamodule, n, lr_of_step=None):
        model.to(self.device).train()
        opt = self._setup_optimizers(model)
        losses = []
        it = iter(datamodule.train_dataloader())
        for i in range(n):
            try:
                batch = next(it)
            except StopIteration:
                it = iter(datamodule.train_dataloader())
                batch = next(it)
            batch = tuple(t.to(self.device) if torch.is_tensor(t) else t for t in batch)
            if lr_of_step is not None:
                for pg in opt.param_groups:
                    pg["lr"] = lr_of_step(i)
            opt.zero_grad(set_to_none=True)
            out = model.training_step(batch, i)
            out["loss"].backward()
            opt.step()
            losses.append(out["loss"].detach().float())
        return torch.stack(losses).cpu().tolist() if losses else []

    def _scale_batch_size(self, model, datamodule, max_trials=6):
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        good = datamodule.batch_size
        for _ in range(max_trials):
            datamodule.batch_size = good * 2
            try:
                self._probe_steps(model, datamodule, 2)
                good = datamodule.batch_size
            except (torch.cuda.OutOfMemoryError, RuntimeError):
                break
        datamodule.batch_size = good
        model.load_state_dict(state)
        return good

    def _lr_find(self, model, datamodule, min_lr=1e-8, max_lr=1.0, num_training=50):
        state = {k: v.detach().clone() for k, v in model.state_dict().items()}
        lrs = [min_lr * (max_lr / min_lr) ** (i / max(num_training - 1, 1)) for i in range(num_training)]
        losses = self._probe_steps(model, datamodule, num_training, lambda i: lrs[i])
        model.load_state_dict(state)
        best, best_slope = None, 0.0
        for i in range(1, len(losses)):
            if losses[i] == losses[i] and losses[i - 1] == losses[i - 1]:
                slope = losses[i] - losses[i - 1]
                if slope < best_slope:
                    best, best_slope = lrs[i], slope
        if best is not None and hasattr(getattr(model, "args", None), "lr"):
            model.args.lr = best
        return best
