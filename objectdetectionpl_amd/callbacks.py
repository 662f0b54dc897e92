"""The callbacks `run.py:88-107` builds, for machines without pytorch_lightning: `ModelCheckpoint`, `EarlyStopping`,
`LearningRateMonitor`, `GPUStatsMonitor` -- same constructor arguments, driven by `objectdetectionpl_amd.trainer.Trainer`
through PL's hook names (`on_fit_start / on_train_epoch_start / on_train_batch_end / on_validation_end / on_fit_end`)."""
import os
import re

import torch


class Callback:
    def on_fit_start(self, trainer, pl_module): pass
    def on_train_epoch_start(self, trainer, pl_module): pass
    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0): pass
    def on_train_epoch_end(self, trainer, pl_module, outputs=None): pass
    def on_validation_end(self, trainer, pl_module): pass
    def on_fit_end(self, trainer, pl_module): pass
    def on_test_end(self, trainer, pl_module): pass


class ModelCheckpoint(Callback):
    """run.py:88-95: monitor / dirpath / filename ('{epoch:02d}-{val_loss:.2f}' -> 'epoch=02-val_loss=0.35') / save_top_k / mode.
    Checkpoints are `{"state_dict", "epoch", "global_step", monitor}` files (PL's key for the weights).  Rank 0 writes."""

    def __init__(self, monitor="val_loss", dirpath=".", filename="model-{epoch:02d}-{val_loss:.2f}", save_top_k=3, mode="min", verbose=False, **_):
        self.monitor, self.dirpath, self.filename, self.save_top_k, self.mode, self.verbose = monitor, dirpath, filename, save_top_k, mode, verbose
        self.best_model_path, self.best_model_score, self._kept = "", None, []

    def _better(self, a, b):
        return a < b if self.mode == "min" else a > b

    def format_name(self, metrics):
        def sub(m):
            name, fmt = m.group(1), m.group(2) or ""
            v = metrics.get(name, 0)
            return "%s=%s" % (name, format(v, fmt) if fmt else str(v))
        return re.sub(r"\{([A-Za-z_][A-Za-z0-9_]*)(?::([^}]*))?\}", sub, self.filename)

    def save_if_better(self, model, epoch, value, global_step=0):
        """Keep the `save_top_k` best epochs; returns the path written or None."""
        if value != value:                                                   # NaN never wins
            return None
        if len(self._kept) >= self.save_top_k > 0 and not self._better(value, self._kept[-1][0]):
            return None
        os.makedirs(self.dirpath, exist_ok=True)
        path = os.path.join(self.dirpath, self.format_name({"epoch": epoch, "step": global_step, self.monitor: value}) + ".ckpt")
        torch.save({"state_dict": model.state_dict(), "epoch": epoch, "global_step": global_step, self.monitor: value}, path)
        self._kept = [kv for kv in self._kept if kv[1] != path] + [(value, path)]
        self._kept.sort(key=lambda t: t[0], reverse=self.mode != "min")
        while self.save_top_k > 0 and len(self._kept) > self.save_top_k:
            _, old = self._kept.pop()
            if os.path.exists(old):
                os.remove(old)
        self.best_model_score, self.best_model_path = self._kept[0]
        if self.verbose:
            print("Epoch %d: %s reached %.5f, saving model to %s" % (epoch, self.monitor, value, path))
        return path

    def on_validation_end(self, trainer, pl_module):
        if self.monitor not in trainer.callback_metrics:
            return
        value = float(trainer.callback_metrics[self.monitor])
        if trainer.is_global_zero:
            self.save_if_better(pl_module, trainer.current_epoch, value, trainer.global_step)
        elif value == value:                                                  # other ranks keep the same bookkeeping, write nothing
            path = os.path.join(self.dirpath, self.format_name({"epoch": trainer.current_epoch, "step": trainer.global_step, self.monitor: value}) + ".ckpt")
            if not (len(self._kept) >= self.save_top_k > 0 and not self._better(value, self._kept[-1][0])):
                self._kept = sorted([kv for kv in self._kept if kv[1] != path] + [(value, path)], key=lambda t: t[0], reverse=self.mode != "min")[:max(self.save_top_k, 1)]
                self.best_model_score, self.best_model_path = self._kept[0]


class EarlyStopping(Callback):
    """run.py:98-104 (monitor='val_loss', min_delta=0.0, patience=3, mode='min'): PL 1.2.6's rule -- an epoch improves when
    `current - min_delta < best` ('min'; mirrored for 'max'); after `patience` epochs without improvement `trainer.should_stop`."""

    def __init__(self, monitor="early_stop_on", min_delta=0.0, patience=3, verbose=False, mode="min", strict=True, **_):
        if mode not in ("min", "max"):
            raise ValueError("mode must be 'min' or 'max'")
        self.monitor, self.patience, self.verbose, self.mode, self.strict = monitor, patience, verbose, mode, strict
        self.min_delta = abs(min_delta) * (1 if mode == "max" else -1)
        self.wait_count, self.stopped_epoch = 0, 0
        self.best_score = float("inf") if mode == "min" else float("-inf")

    def _improved(self, current):
        return (current - self.min_delta) > self.best_score if self.mode == "max" else (current - self.min_delta) < self.best_score

    def on_validation_end(self, trainer, pl_module):
        if getattr(trainer, "running_sanity_check", False):
            return
        if self.monitor not in trainer.callback_metrics:
            if self.strict:
                raise RuntimeError(f"Early stopping conditioned on metric `{self.monitor}` which is not available. "
                                   f"Available metrics are: {sorted(trainer.callback_metrics)}")
            return
        current = float(trainer.callback_metrics[self.monitor])
        if self._improved(current):
            self.best_score, self.wait_count = current, 0
        else:
            self.wait_count += 1
            if self.wait_count >= self.patience:
                self.stopped_epoch = trainer.current_epoch
                trainer.should_stop = True
                if self.verbose:
                    print("Epoch %d: early stopping (%s did not improve for %d epochs)" % (trainer.current_epoch, self.monitor, self.patience))


class LearningRateMonitor(Callback):
    """run.py:106: logging_interval='epoch' (or 'step'): the learning rate of every optimizer / param group goes to the logger
    as `lr-<OptimizerClass>[/pg<i>]`."""

    def __init__(self, logging_interval=None, log_momentum=False):
        if logging_interval not in (None, "step", "epoch"):
            raise ValueError("logging_interval should be `step` or `epoch` or `None`.")
        self.logging_interval, self.log_momentum = logging_interval, log_momentum
        self.lrs = {}

    def _extract(self, trainer):
        out = {}
        for oi, opt in enumerate(trainer.optimizers):
            base = "lr-" + type(opt).__name__ + ("" if oi == 0 else "-%d" % oi)
            for gi, pg in enumerate(opt.param_groups):
                out[base if len(opt.param_groups) == 1 else "%s/pg%d" % (base, gi + 1)] = pg["lr"]
        for k, v in out.items():
            self.lrs.setdefault(k, []).append(v)
        if trainer.logger is not None and out:
            trainer.logger.log_metrics(out, step=trainer.global_step)
        return out

    def on_train_epoch_start(self, trainer, pl_module):
        if self.logging_interval != "step":
            self._extract(trainer)

    def on_train_batch_end(self, trainer, pl_module, outputs, batch, batch_idx, dataloader_idx=0):
        if self.logging_interval == "step":
            self._extract(trainer)


class GPUStatsMonitor(Callback):
    """run.py:107.  PL 1.2.6 shells out to `nvidia-smi` every step; on ROCm the same figures come from the runtime without
    leaving the process (no host sync: allocator counters only) and are logged once per epoch."""

    def __init__(self, memory_utilization=True, gpu_utilization=True, intra_step_time=False, inter_step_time=False,
                 fan_speed=False, temperature=False):
        self.memory_utilization = memory_utilization
        self.stats = []

    def on_train_epoch_end(self, trainer, pl_module, outputs=None):
        dev = trainer.device
        if dev.type != "cuda" or not self.memory_utilization:
            return
        rec = {"gpu_id: %d/memory.used (MB)" % (dev.index or 0): torch.cuda.memory_allocated(dev) / 2 ** 20,
               "gpu_id: %d/memory.peak (MB)" % (dev.index or 0): torch.cuda.max_memory_allocated(dev) / 2 ** 20}
        self.stats.append(rec)
        if trainer.logger is not None:
            trainer.logger.log_metrics(rec, step=trainer.global_step)
