"""In-repo stand-in for the `pytorch_lightning` names the reference uses (run.py:1-4; model/*.py `pl.LightningModule`;
dataset/*.py `pl.LightningDataModule`).  Own code over `objectdetectionpl_amd.trainer` -- not PL, not a copy of it."""
import torch
import torch.nn as nn

from objectdetectionpl_amd.trainer import Trainer  # noqa: F401

from . import callbacks, loggers  # noqa: F401

__version__ = "0.0-odpl-standin"


def seed_everything(seed=None):
    import random
    seed = 0 if seed is None else int(seed)
    random.seed(seed)
    torch.manual_seed(seed)
    try:
        import numpy as np
        np.random.seed(seed % (2 ** 32))
    except ImportError:
        pass
    return seed


class LightningModule(nn.Module):
    """The attributes and hooks the reference's models and step functions touch: `self.logger.experiment.add_*`,
    `self.log_dict`, `self.global_step`, `self.current_epoch` (step.py:21-62), `load_from_checkpoint`
    (lightningUtils.py:32), `self.trainer`."""

    def __init__(self, *a, **k):
        super().__init__()
        self.__dict__["trainer"] = None
        self.__dict__["_logger"] = None
        self.__dict__["_progress"] = {"current_epoch": 0, "global_step": 0}
        self.__dict__["_logged_metrics"] = {}

    # progress: the trainer's counters when attached, else what was last set
    @property
    def current_epoch(self):
        return self.trainer.current_epoch if self.trainer is not None else self._progress["current_epoch"]

    @current_epoch.setter
    def current_epoch(self, v):
        self._progress["current_epoch"] = v

    @property
    def global_step(self):
        return self.trainer.global_step if self.trainer is not None else self._progress["global_step"]

    @global_step.setter
    def global_step(self, v):
        self._progress["global_step"] = v

    @property
    def logger(self):
        if self._logger is not None:
            return self._logger
        return self.trainer.logger if self.trainer is not None else None

    @logger.setter
    def logger(self, v):
        self.__dict__["_logger"] = v

    @property
    def device(self):
        p = next(self.parameters(), None)
        return p.device if p is not None else torch.device("cpu")

    def log(self, name, value, prog_bar=False, logger=True, on_step=None, on_epoch=None, **_):
        self._logged_metrics[name] = value.detach() if torch.is_tensor(value) else value
        if self.trainer is not None:
            self.trainer.callback_metrics[name] = self._logged_metrics[name]

    def log_dict(self, dictionary, **kw):
        for k, v in dictionary.items():
            self.log(k, v, **kw)

    @classmethod
    def load_from_checkpoint(cls, checkpoint_path, map_location=None, strict=True, **kwargs):
        """`cls(**kwargs)` with the checkpoint's `state_dict` loaded (weights only: nothing from the file is executed)."""
        ck = torch.load(checkpoint_path, map_location=map_location or "cpu", weights_only=True)
        model = cls(**kwargs)
        model.load_state_dict(ck["state_dict"], strict=strict)
        return model

    def freeze(self):
        for p in self.parameters():
            p.requires_grad = False
        self.eval()

    def unfreeze(self):
        for p in self.parameters():
            p.requires_grad = True
        self.train()

    def summarize(self, mode="top"):
        return "\n".join("%s: %d" % (n, sum(p.numel() for p in m.parameters())) for n, m in self.named_children())


class LightningDataModule:
    """`setup(stage)` / `*_dataloader()` container (dataset/Pascal.py:190-260).  `setup` runs once per stage however often it
    is called (run.py:47 calls it, then the trainer does) -- PL's `has_setup_fit` / `has_setup_test` bookkeeping."""
    name = ""

    def __init__(self, *a, **k):
        self._has_setup = set()

    def __init_subclass__(cls, **kw):
        super().__init_subclass__(**kw)
        raw = cls.__dict__.get("setup")
        if raw is not None:
            def setup(self, stage=None, _raw=raw):
                done = self.__dict__.setdefault("_has_setup", set())
                if stage in done:
                    return None
                done.add(stage)
                return _raw(self, stage)
            cls.setup = setup

    @property
    def has_setup_fit(self):
        return "fit" in self.__dict__.get("_has_setup", ())

    @property
    def has_setup_test(self):
        return "test" in self.__dict__.get("_has_setup", ())

    def prepare_data(self):
        pass

    def setup(self, stage=None):
        pass
