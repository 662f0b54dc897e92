"""Scalar sink with the TensorBoardLogger surface `run.py:86` and `LightningFunc/step.py:23-158` use.

tensorboard is not installed here (and the reference's logging is out of scope as a product: SURVEY section 2 row 15); what the
hot path needs is a place to put the scalars it produces WITHOUT a host sync per value.  `experiment` mimics the
`SummaryWriter` calls the step functions make (`add_scalars / add_scalar / add_histogram / add_graph / add_image`) and appends
JSON lines to `<save_dir>/<name>/version_<k>/scalars.jsonl`; histograms are reduced to (count, mean, std, min, max)."""
import json
import os
import time


class ScalarWriter:
    def __init__(self, log_dir):
        self.log_dir = log_dir
        self._f = None
        self.counts = {"scalars": 0, "histograms": 0, "graphs": 0, "images": 0}

    def _write(self, rec):
        if self._f is None:
            os.makedirs(self.log_dir, exist_ok=True)
            self._f = open(os.path.join(self.log_dir, "scalars.jsonl"), "a", encoding="utf-8")
        rec["wall"] = round(time.time(), 3)
        self._f.write(json.dumps(rec) + "\n")

    @staticmethod
    def _num(v):
        return float(v.item()) if hasattr(v, "item") else float(v)

    def add_scalar(self, tag, value, global_step=None, **_):
        self.counts["scalars"] += 1
        self._write({"tag": tag, "value": self._num(value), "step": None if global_step is None else int(global_step)})

    def add_scalars(self, main_tag, tag_scalar_dict, global_step=None, **_):
        for k, v in tag_scalar_dict.items():
            self.counts["scalars"] += 1
            self._write({"tag": main_tag, "series": k, "value": self._num(v), "step": None if global_step is None else int(global_step)})

    def add_histogram(self, tag, values, global_step=None, **_):
        import torch
        self.counts["histograms"] += 1
        t = values.detach().float().flatten() if torch.is_tensor(values) else torch.as_tensor(values, dtype=torch.float32).flatten()
        if t.numel():
            st = torch.stack([t.mean(), t.std(unbiased=False), t.min(), t.max()]).tolist()      # one device->host copy per histogram
            self._write({"tag": tag, "hist": {"count": t.numel(), "mean": st[0], "std": st[1], "min": st[2], "max": st[3]},
                         "step": None if global_step is None else int(global_step)})

    def add_graph(self, *_, **__):
        self.counts["graphs"] += 1

    def add_image(self, *_, **__):
        self.counts["images"] += 1

    add_images = add_image

    def add_text(self, tag, text, global_step=None, **_):
        self._write({"tag": tag, "text": str(text), "step": None if global_step is None else int(global_step)})

    def flush(self):
        if self._f is not None:
            self._f.flush()

    def close(self):
        if self._f is not None:
            self._f.close()
            self._f = None


class _NoWriter:
    """PL's DummyExperiment: what `logger.experiment` is on every rank but 0 -- accepts any SummaryWriter call, writes nothing."""

    def __getattr__(self, _name):
        return lambda *a, **k: None


def _rank_zero():
    return int(os.environ.get("RANK", os.environ.get("LOCAL_RANK", "0")) or 0) == 0


class TensorBoardLogger:
    """`TensorBoardLogger(save_dir, name=..., version=None, default_hp_metric=False)` (run.py:86).  Only rank 0 writes (PL's
    `rank_zero_experiment`): on the other ranks of a data-parallel job `experiment` is a sink and no `version_k` directory is made."""

    def __init__(self, save_dir, name="default", version=None, default_hp_metric=True, **_):
        self.save_dir, self.name = save_dir, name or ""
        self._version = version
        self._experiment = None

    @property
    def root_dir(self):
        return os.path.join(self.save_dir, self.name)

    @property
    def version(self):
        if self._version is None:
            existing = []
            if os.path.isdir(self.root_dir):
                for d in os.listdir(self.root_dir):
                    if d.startswith("version_") and d[8:].isdigit():
                        existing.append(int(d[8:]))
            self._version = max(existing) + 1 if existing else 0
        return self._version

    @property
    def log_dir(self):
        v = self.version
        return os.path.join(self.root_dir, v if isinstance(v, str) else f"version_{v}")

    @property
    def experiment(self):
        if not _rank_zero():
            return _NoWriter()
        if self._experiment is None:
            self._experiment = ScalarWriter(self.log_dir)
        return self._experiment

    def log_metrics(self, metrics, step=None):
        for k, v in metrics.items():
            self.experiment.add_scalar(k, v, step)

    def log_hyperparams(self, params=None, metrics=None):
        if params is not None:
            d = vars(params) if hasattr(params, "__dict__") else dict(params)
            self.experiment.add_text("hparams", json.dumps({k: repr(v) for k, v in d.items()}))

    def save(self):
        self.experiment.flush()

    def finalize(self, status="success"):
        self.experiment.flush()
        self.experiment.close()
