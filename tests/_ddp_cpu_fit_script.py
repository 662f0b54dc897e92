"""Run by tests/test_trainer.py: ONE process start -> Trainer(accelerator="ddp_cpu", num_processes=2) launches rank 1 itself
(the reference's multi-GPU entry is one `python run.py`, configs/config.yaml:58-61), both ranks fit a tiny BN network on
different data with gradient accumulation, and each writes what it ended with to <out>/rank<r>.pt."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn as nn

from objectdetectionpl_amd.trainer import Trainer

out_dir = sys.argv[1]
mode = sys.argv[2] if len(sys.argv) > 2 else "gen"      # gen: per-rank generators | loader: DataLoader datamodule | fail: rank 1 dies


class Tiny(nn.Module):
    def __init__(self):
        super().__init__()
        self.conv = nn.Conv2d(3, 8, 3, padding=1)
        self.bn = nn.BatchNorm2d(8)
        self.fc = nn.Linear(8, 1)
        self.steps = 0

    def forward(self, x):
        return self.fc(torch.relu(self.bn(self.conv(x))).mean((2, 3)))

    def training_step(self, batch, i):
        x, y = batch
        return {"loss": ((self.forward(x) - y) ** 2).mean()}

    def validation_step(self, batch, i):
        x, y = batch
        return {"loss": ((self.forward(x) - y) ** 2).mean()}

    def training_epoch_end(self, outs):
        return []

    def validation_epoch_end(self, outs):
        return []

    def configure_optimizers(self):
        opt = torch.optim.SGD(self.parameters(), lr=0.1)
        real_step = opt.step

        def counted(*a, **k):
            self.steps += 1
            return real_step(*a, **k)
        opt.step = counted
        # a plateau scheduler that fires as soon as val_loss fails to improve: rank-local val losses would desynchronise it
        return {"optimizer": opt, "lr_scheduler": torch.optim.lr_scheduler.ReduceLROnPlateau(opt, mode="min", patience=0, factor=0.5, threshold=0.5),
                "monitor": "val_loss"}


class Data:
    name = "tiny"

    def setup(self, stage=None):
        pass

    def _gen(self, seed, n):
        g = torch.Generator().manual_seed(seed + 1000 * int(os.environ.get("RANK", "0")))     # different data per rank
        for _ in range(n):
            x = torch.randn(4, 3, 8, 8, generator=g)
            yield x, x.mean((1, 2, 3), keepdim=False)[:, None] * 3 + int(os.environ.get("RANK", "0"))

    def train_dataloader(self):
        return self._gen(1, 5)                      # 5 batches, accumulate 2 -> optimiser steps after batches 2, 4 and 5 (the last)

    def val_dataloader(self):
        return self._gen(2, 3)


class _Indexed(torch.utils.data.Dataset):
    """item i -> (x_i, y_i, i): the index travels with the sample so that each rank can report what it was given"""

    def __init__(self, n, seed):
        g = torch.Generator().manual_seed(seed)
        self.x = torch.randn(n, 3, 8, 8, generator=g)
        self.y = self.x.mean((1, 2, 3))[:, None] * 3

    def __len__(self):
        return self.x.shape[0]

    def __getitem__(self, i):
        return self.x[i], self.y[i], torch.tensor(i)


class LoaderData:
    """what run.py / dataset/Pascal.py build: plain DataLoaders, shuffle=True for training, no sampler of their own.  The
    validation side is deliberately UNEVEN (a generator: rank 0 gets three batches, rank 1 one) -- per-forward collectives would
    deadlock on it."""
    name = "loader"

    def setup(self, stage=None):
        self.train = _Indexed(22, 5)

    def train_dataloader(self):
        return torch.utils.data.DataLoader(self.train, batch_size=4, shuffle=True, generator=torch.Generator().manual_seed(7))

    def val_dataloader(self):
        g = torch.Generator().manual_seed(3)
        for _ in range(3 if int(os.environ.get("RANK", "0")) == 0 else 1):
            x = torch.randn(4, 3, 8, 8, generator=g)
            yield x, x.mean((1, 2, 3))[:, None] * 3, torch.zeros(4)


class TinyIdx(Tiny):
    def __init__(self):
        super().__init__()
        self.seen = []            # per epoch: indices of the training samples this rank was given

    def training_step(self, batch, i):
        x, y, idx = batch
        if i == 0:
            self.seen.append([])
        self.seen[-1] += idx.tolist()
        if self.logger is not None:
            self.logger.experiment.add_scalar("train_loss", 0.0, i)
        return {"loss": ((self.forward(x) - y) ** 2).mean()}

    def validation_step(self, batch, i):
        x, y, _ = batch
        return {"loss": ((self.forward(x) - y) ** 2).mean()}


torch.manual_seed(int(os.environ.get("RANK", "0")))                                            # different initial weights per rank
if mode == "fail":
    if os.environ.get("RANK") == "1":
        sys.exit(7)                                     # a rank that dies at start-up (LOCAL_RANK >= device_count, an import error, ...)
    model = Tiny()
    trainer = Trainer(max_epochs=3, accelerator="ddp_cpu", num_processes=2, num_sanity_val_steps=0)
    trainer.fit(model, Data())                          # rank 0 waits in the rendezvous / first broadcast: the watcher ends the job
    sys.exit(0)
if mode == "loader":
    from objectdetectionpl_amd.loggers import TensorBoardLogger
    model = TinyIdx()
    logger = TensorBoardLogger(os.path.join(out_dir, "tb"), name="tiny")
    trainer = Trainer(max_epochs=2, accelerator="ddp_cpu", num_processes=2, num_sanity_val_steps=0, logger=logger)
    model.logger = logger
    trainer.fit(model, LoaderData())
    torch.save({"seen": model.seen, "val_loss": float(trainer.callback_metrics["val_loss"]), "rank": trainer.rank,
                "params": {k: v.clone() for k, v in model.state_dict().items()},
                "group_alive": torch.distributed.is_initialized()}, os.path.join(out_dir, "rank%d.pt" % trainer.rank))
    sys.exit(0)
model = Tiny()
trainer = Trainer(max_epochs=3, accumulate_grad_batches=2, accelerator="ddp_cpu", num_processes=2, num_sanity_val_steps=0)
trainer.fit(model, Data())                              # fit ends with a barrier and destroys the process group it created
torch.save({"params": {k: v.clone() for k, v in model.state_dict().items()}, "lr": trainer.optimizers[0].param_groups[0]["lr"],
            "steps": model.steps, "val_loss": float(trainer.callback_metrics["val_loss"]), "world": trainer.world, "rank": trainer.rank},
           os.path.join(out_dir, "rank%d.pt" % trainer.rank))
