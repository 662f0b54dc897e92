"""The PL-shaped host around the hot path (`trainer.py`, `callbacks.py`, `loggers.py`, `compat/pytorch_lightning`): run.py:86-118's
objects without pytorch_lightning.  CPU only; the device flow is tests/test_gpu_end_to_end.py."""
import argparse
import json
import os
import subprocess
import sys

import pytest
import torch
import torch.nn as nn

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_fit_world2_one_process_start_keeps_ranks_identical(tmp_path):
    """ADVICE r01 (high + medium): `Trainer.fit` at world 2 -- started as ONE process, which launches rank 1 itself -- trains
    with accumulation (last partial group stepped), averages gradients, re-broadcasts buffers, reduces val_loss over the ranks
    before the plateau scheduler: parameters, buffers and learning rates end identical on both ranks."""
    r = _run_ddp_script(tmp_path, "gen")
    assert r.returncode == 0, r.stderr[-3000:]
    a = torch.load(tmp_path / "rank0.pt", weights_only=True)
    b = torch.load(tmp_path / "rank1.pt", weights_only=True)
    assert a["world"] == b["world"] == 2 and (a["rank"], b["rank"]) == (0, 1)
    assert a["steps"] == b["steps"] == 3 * 3                         # ceil(5 / 2) optimiser steps per epoch, 3 epochs
    assert a["lr"] == b["lr"] and a["lr"] < 0.1                      # the plateau scheduler fired, identically
    assert a["val_loss"] == b["val_loss"]                            # the rank-averaged figure
    for k in a["params"]:
        assert torch.equal(a["params"][k], b["params"][k]), k        # weights AND BatchNorm buffers


def _run_ddp_script(tmp_path, mode, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(REPO, "tests", "_ddp_cpu_fit_script.py"), str(tmp_path), mode], env=env,
                          capture_output=True, text=True, timeout=timeout)


def test_fit_world2_shards_a_dataloader_and_survives_uneven_validation(tmp_path):
    """a datamodule that returns plain DataLoaders (run.py, dataset/Pascal.py) is sharded like PL's
    `replace_sampler_ddp=True` -- the ranks' index sets are disjoint, together they cover the dataset, the shuffle changes with the
    epoch; validation with a different number of batches per rank finishes (no collective inside a validation forward) and both
    ranks decide on the same val_loss; only rank 0 writes logs; fit leaves no process group behind."""
    r = _run_ddp_script(tmp_path, "loader")
    assert r.returncode == 0, r.stderr[-3000:]
    a = torch.load(tmp_path / "rank0.pt", weights_only=True)
    b = torch.load(tmp_path / "rank1.pt", weights_only=True)
    for ep in range(2):
        sa, sb = set(a["seen"][ep]), set(b["seen"][ep])
        assert len(a["seen"][ep]) == len(b["seen"][ep]) == 11              # 22 samples over two ranks
        assert not (sa & sb) and (sa | sb) == set(range(22))
    assert a["seen"][0] != a["seen"][1]                                    # set_epoch: another permutation in epoch 1
    assert a["val_loss"] == b["val_loss"]
    for k in a["params"]:
        assert torch.equal(a["params"][k], b["params"][k]), k
    assert not a["group_alive"] and not b["group_alive"]
    versions = sorted(os.listdir(tmp_path / "tb" / "tiny"))
    assert versions == ["version_0"]                                       # one writer
    tags = [json.loads(l)["tag"] for l in open(tmp_path / "tb" / "tiny" / "version_0" / "scalars.jsonl")]
    assert tags.count("val_loss") == 2 and tags.count("train_loss") == 2 * 3      # rank 0's three training batches per epoch, once


def test_a_dying_rank_takes_the_job_down_promptly(tmp_path):
    """rank 1 exits with code 7 before the rendezvous; rank 0 (the process the user started) must not wait
    in a collective: the job ends non-zero within seconds and no child is left behind."""
    import time
    t0 = time.time()
    r = _run_ddp_script(tmp_path, "fail", timeout=120)
    assert r.returncode == 7, (r.returncode, r.stderr[-2000:])
    assert time.time() - t0 < 60
    assert "rank 1 exited with code 7" in r.stderr


class _Tiny(nn.Module):
    def __init__(self, val_losses):
        super().__init__()
        self.w = nn.Parameter(torch.zeros(1))
        self.val_losses = list(val_losses)
        self.epochs_seen = 0

    def forward(self, x):
        return x * self.w

    def training_step(self, batch, i):
        (x,) = batch
        return {"loss": ((self.forward(x) - 1.0) ** 2).mean()}

    def validation_step(self, batch, i):
        return {"loss": torch.tensor(self.val_losses[min(self.epochs_seen, len(self.val_losses) - 1)])}

    def validation_epoch_end(self, outs):
        self.epochs_seen += 1
        return []


class _Data:
    name = "d"

    def __init__(self):
        self.setups = []

    def setup(self, stage=None):
        self.setups.append(stage)

    def train_dataloader(self):
        return [(torch.ones(2),)] * 3

    val_dataloader = test_dataloader = train_dataloader


def test_early_stopping_checkpoints_lr_monitor_and_logger(tmp_path):
    from objectdetectionpl_amd.callbacks import EarlyStopping, GPUStatsMonitor, LearningRateMonitor, ModelCheckpoint
    from objectdetectionpl_amd.loggers import TensorBoardLogger
    from objectdetectionpl_amd.trainer import Trainer
    # val_loss improves twice, then stalls: patience 3 (run.py:98-104) stops after epoch index 4
    model = _Tiny([1.0, 0.5, 0.6, 0.7, 0.55, 0.1, 0.1])
    ck = ModelCheckpoint(monitor="val_loss", dirpath=str(tmp_path / "ck"), filename="m-{epoch:02d}-{val_loss:.2f}", save_top_k=2, mode="min")
    es = EarlyStopping(monitor="val_loss", min_delta=0.0, patience=3, mode="min")
    lrm = LearningRateMonitor(logging_interval="epoch")
    logger = TensorBoardLogger(str(tmp_path / "tb"), name="Tiny", default_hp_metric=False)
    cfg = argparse.Namespace(max_epochs=50, accumulate_grad_batches=2, num_sanity_val_steps=0, gpus=0, auto_select_gpus=True,
                             some_unrelated_yaml_key=7)
    tr = Trainer.from_argparse_args(cfg, callbacks=[ck, es, lrm, GPUStatsMonitor()], logger=logger, device="cpu")
    tr.fit(model, datamodule=_Data())
    assert tr.should_stop and tr.current_epoch == 4 and es.stopped_epoch == 4 and es.best_score == 0.5
    kept = sorted(os.listdir(tmp_path / "ck"))
    assert kept == ["m-epoch=01-val_loss=0.50.ckpt", "m-epoch=04-val_loss=0.55.ckpt"]           # top-2 of {1.0, .5, .6, .7, .55}
    assert ck.best_model_path.endswith("m-epoch=01-val_loss=0.50.ckpt") and ck.best_model_score == 0.5
    assert model.checkpoint_callback is ck                                                   # run.py:96 setattr, done for the caller
    assert len(lrm.lrs["lr-Adam"]) == 5
    lines = [json.loads(l) for l in open(os.path.join(logger.log_dir, "scalars.jsonl"))]
    assert sum(1 for l in lines if l["tag"] == "val_loss") == 5 and any(l["tag"] == "lr-Adam" for l in lines)
    ckpt = torch.load(ck.best_model_path, weights_only=True)
    assert set(ckpt) >= {"state_dict", "epoch", "val_loss"} and ckpt["epoch"] == 1


def test_setup_is_called_per_stage_and_test_runs():
    from objectdetectionpl_amd.trainer import Trainer
    model = _Tiny([0.3])
    model.test_step = lambda b, i: {"n": 1}
    model.test_epoch_end = lambda outs: {"batches": len(outs)}
    dm = _Data()
    tr = Trainer(max_epochs=1, limit_test_batches=2, num_sanity_val_steps=1, device="cpu")
    tr.fit(model, dm)
    assert tr.test(model, datamodule=dm) == {"batches": 2}
    assert dm.setups == ["fit", "test"]


def test_grad_averager_resets_when_a_bucket_does_not_fill():
    """a parameter without a gradient raises once, and the exchange state is clean for the next step."""
    import torch.distributed as dist
    from objectdetectionpl_amd.parallel import GradAverager
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        a, b = nn.Parameter(torch.ones(3)), nn.Parameter(torch.ones(3))
        avg = GradAverager([a, b], bucket_mb=1, first_bucket_mb=1)
        with pytest.raises(RuntimeError, match="did not fill"):
            (a * 2).sum().backward()                       # b gets no gradient
        assert not avg._armed and all(bk.pending == len(bk.params) and bk.work is None for bk in avg.buckets)
        a.grad = b.grad = None
        (a * 2 + b * 3).sum().backward()                   # the next, complete step works
        assert torch.equal(a.grad, torch.full((3,), 2.0)) and torch.equal(b.grad, torch.full((3,), 3.0))
    finally:
        dist.destroy_process_group()


def test_pytorch_lightning_names_resolve_to_the_stand_in(tmp_path):
    """run.py:1-4's imports, in a fresh interpreter, with only objectdetectionpl_amd.compat on the path."""
    code = """
import sys
sys.path.insert(0, %r)
from objectdetectionpl_amd import compat
pl = compat.install()
from pytorch_lightning import Trainer
from pytorch_lightning.loggers import TensorBoardLogger
from pytorch_lightning.callbacks import ModelCheckpoint, LearningRateMonitor, GPUStatsMonitor
from pytorch_lightning.callbacks.early_stopping import EarlyStopping
import pytorch_lightning
class DM(pytorch_lightning.LightningDataModule):
    def __init__(self):
        super().__init__(); self.n = 0
    def setup(self, stage=None):
        self.n += 1
d = DM(); d.setup('fit'); d.setup('fit'); d.setup('test')
class M(pytorch_lightning.LightningModule):
    def __init__(self, k=1):
        super().__init__(); self.l = __import__('torch').nn.Linear(k, 1)
m = M(); m.log_dict({'val_loss': 1.5}); m.global_step = 3
import torch
torch.save({'state_dict': M(2).state_dict()}, %r)
m2 = M.load_from_checkpoint(%r, k=2)
print(d.n, d.has_setup_fit, m.global_step, m.current_epoch, m.logger, m2.l.in_features, hasattr(Trainer, 'from_argparse_args'), 'standin' in pytorch_lightning.__version__)
""" % (REPO, str(tmp_path / "m.ckpt"), str(tmp_path / "m.ckpt"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split() == ["2", "True", "3", "0", "None", "2", "True", "True"]


def test_forward_via_offers_a_folded_step_to_the_bucketed_wrapper_only():
    """`_ForwardVia.forward_and_loss`: the model's fold runs behind `DataParallelHost.sync_buffers()`; torch's DDP (which must see
    `forward`) and a model that declines get None, i.e. `training_step` goes forward + criterion through the wrapper."""
    from objectdetectionpl_amd import parallel
    from objectdetectionpl_amd.trainer import _ForwardVia
    calls = []

    class M:
        take = True

        def loss_driven_mask(self, x, y):
            return 1 if self.take else 0

        def forward_and_loss(self, x, y, mask=None):
            calls.append(("fold", mask))
            return {"loss": x + y}

    host = object.__new__(parallel.DataParallelHost)
    host.sync_buffers = lambda: calls.append("sync")
    m = M()
    via = _ForwardVia(m, host)
    assert via.forward_and_loss(1, 2) == {"loss": 3} and calls == ["sync", ("fold", 1)]      # the decision is handed down, not re-made
    m.take = False
    del calls[:]
    assert via.forward_and_loss(1, 2) is None and calls == []

    # the route decision is not pure (model/RetinaNet.py:_crowded_levels polls an asynchronous read-back): a second look inside the
    # fold could disagree with the first and make this rank broadcast its buffers twice in one step.  One look per step:
    class Flipping(M):
        looks = 0

        def loss_driven_mask(self, x, y):
            self.looks += 1
            return 3 if self.looks == 1 else 0            # every later look would decline

        def forward_and_loss(self, x, y, mask=None):
            if mask is None:
                mask = self.loss_driven_mask(x, y)
            calls.append(("fold", mask))
            return {"loss": x + y} if mask else None
    f = Flipping()
    del calls[:]
    assert _ForwardVia(f, host).forward_and_loss(1, 2) == {"loss": 3}
    assert calls == ["sync", ("fold", 3)] and f.looks == 1                                   # exactly one sync_buffers, one look
    assert _ForwardVia(m, object()).forward_and_loss is None                 # any other wrapper (torch DDP)

    class Plain:
        pass
    assert _ForwardVia(Plain(), host).forward_and_loss is None


def test_benchmark_flag_turns_the_solver_search_on():
    """PL's Trainer(benchmark=True) == torch.backends.cudnn.benchmark (MIOpen's solver search on ROCm); the default leaves it alone"""
    import torch
    from objectdetectionpl_amd.trainer import Trainer
    old = torch.backends.cudnn.benchmark
    try:
        torch.backends.cudnn.benchmark = False
        Trainer(max_epochs=1, device="cpu")
        assert torch.backends.cudnn.benchmark is False
        assert Trainer(max_epochs=1, device="cpu", benchmark=True).benchmark and torch.backends.cudnn.benchmark is True
    finally:
        torch.backends.cudnn.benchmark = old
