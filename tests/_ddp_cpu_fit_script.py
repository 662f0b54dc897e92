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


torch.manual_seed(int(os.environ.get("RANK", "0")))                                            # different initial weights per rank
model = Tiny()
trainer = Trainer(max_epochs=3, accumulate_grad_batches=2, accelerator="ddp_cpu", num_processes=2, num_sanity_val_steps=0)
trainer.fit(model, Data())
torch.save({"params": {k: v.clone() for k, v in model.state_dict().items()}, "lr": trainer.optimizers[0].param_groups[0]["lr"],
            "steps": model.steps, "val_loss": float(trainer.callback_metrics["val_loss"]), "world": trainer.world, "rank": trainer.rank},
           os.path.join(out_dir, "rank%d.pt" % trainer.rank))
import torch.distributed as dist
dist.barrier()
dist.destroy_process_group()
