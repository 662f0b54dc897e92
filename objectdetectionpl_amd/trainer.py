"""A minimal stand-in for the subset of `pytorch_lightning.Trainer` that `run.py:109-118` uses (fit / test with
max_epochs, limit_*_batches, accumulate_grad_batches), for machines without PL.  One process per GPU; with
WORLD_SIZE > 1 the module is wrapped by `parallel.wrap_data_parallel` (bucketed gradient averaging over RCCL) and
micro-steps that do not end in an optimiser step run under `no_sync`.  `callbacks` may hold a
`LightningFunc.lightningUtils.ModelCheckpoint` (run.py:86-93): rank 0 writes the top-k `val_loss` checkpoints and the
model's `write_Best_model_path` records the best one.  Not a training framework: early stopping, LR monitors and
TensorBoard stay with PL (out of scope)."""
import contextlib
import os

import torch

from . import parallel


class Trainer:
    def __init__(self, max_epochs=1, limit_train_batches=None, limit_val_batches=None, limit_test_batches=None,
                 accumulate_grad_batches=1, device=None, callbacks=(), **_ignored):
        self.max_epochs = max_epochs
        self.limits = dict(train=limit_train_batches, val=limit_val_batches, test=limit_test_batches)
        self.accumulate = max(1, int(accumulate_grad_batches))
        self.device = torch.device(device or ("cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.checkpoint_callback = next((c for c in callbacks if hasattr(c, "on_validation_end")), None)

    @classmethod
    def from_argparse_args(cls, args, **kw):
        keys = ("max_epochs", "limit_train_batches", "limit_val_batches", "limit_test_batches", "accumulate_grad_batches")
        return cls(**{k: getattr(args, k) for k in keys if hasattr(args, k)}, **kw)

    def _take(self, loader, kind):
        lim = self.limits[kind]
        for i, batch in enumerate(loader):
            if lim is not None and i >= lim:
                break
            yield i, tuple(t.to(self.device, non_blocking=True) for t in batch)

    def _optimizer(self, model):
        """-> (optimizer, scheduler or None, monitor).  A model carrying `args.optimizer` uses the reference's factory
        (`LightningFunc/optimizer.py:configure_optimizers`); otherwise fused Adam on `args.lr` / `args.weight_decay`."""
        a = model.args
        if hasattr(model, "configure_optimizers") and hasattr(a, "optimizer") and hasattr(a, "lr_scheduler"):
            cfg = model.configure_optimizers()
            return cfg["optimizer"], cfg.get("lr_scheduler"), cfg.get("monitor", "val_loss")
        return (torch.optim.Adam(model.parameters(), lr=getattr(a, "lr", 1e-3), weight_decay=getattr(a, "weight_decay", 0.0), fused=True),
                None, "val_loss")

    def fit(self, model, datamodule):
        model.to(self.device).train()
        if self.world > 1:
            parallel.init_process_group_from_env(self.device)
        net = parallel.wrap_data_parallel(model, self.device) if self.world > 1 else model
        opt, sched, _monitor = self._optimizer(model)
        history = []
        model.global_step = 0
        for epoch in range(self.max_epochs):
            model.current_epoch = epoch
            model.train()
            outs = []
            for i, (x, y) in self._take(datamodule.train_dataloader(), "train"):
                fwd = model.forward
                model.forward = net if net is not model else fwd            # route forward through DDP
                try:
                    out = model.training_step((x, y), i)
                finally:
                    model.forward = fwd
                last_micro = (i + 1) % self.accumulate == 0
                with (contextlib.nullcontext() if last_micro or not hasattr(net, "no_sync") else net.no_sync()):
                    (out["loss"] / self.accumulate).backward()
                if (i + 1) % self.accumulate == 0:
                    opt.step()
                    opt.zero_grad(set_to_none=True)
                model.global_step += 1
                outs.append({k: v.detach() for k, v in out.items()})
            if outs:
                history.append(("train", epoch, model.training_epoch_end(outs)))
            model.eval()
            vouts = [model.validation_step(b, i) for i, b in self._take(datamodule.val_dataloader(), "val")]
            if vouts and sched is not None:              # epoch-level scheduler step, the plateau one on val_loss (PL's "monitor")
                if isinstance(sched, torch.optim.lr_scheduler.ReduceLROnPlateau):
                    sched.step(float(torch.stack([o["loss"].detach() for o in vouts]).mean()))
                else:
                    sched.step()
            if vouts:
                if self.checkpoint_callback is not None and self.rank == 0:
                    model.checkpoint_callback = self.checkpoint_callback
                    val_loss = float(torch.stack([o["loss"].detach() for o in vouts]).mean())
                    self.checkpoint_callback.on_validation_end(model, epoch, val_loss)
                history.append(("val", epoch, model.validation_epoch_end(vouts)))
        return history

    def test(self, model, datamodule):
        model.to(self.device).eval()
        outs = [model.test_step(b, i) for i, b in self._take(datamodule.test_dataloader(), "test")]
        return model.test_epoch_end(outs) if outs else {}
