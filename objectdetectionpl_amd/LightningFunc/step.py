"""Step glue: mirror of `LightningFunc/step.py:14-158` (the functions the reference setattr's onto every model
class) without its per-key host synchronisations.

The reference logs each of the 7 loss scalars with `logger.experiment.add_scalars(...)` every training step
(step.py:21-23): 7 device->host syncs per step.  Here scalars stay on the device in a `DeferredScalars` ring and are
flushed with ONE copy every `flush_every` steps (or at epoch end).  cv2 overlays (step.py:72-90) are visualisation:
out of scope.  A model without a logger simply skips logging."""
import numpy as np
import torch

from .accuracy import ap_per_class, get_batch_statistics


class DeferredScalars:
    """Collects (tag, series, device scalar, step) and writes them to a TensorBoard-like sink in batches."""

    def __init__(self, sink=None, flush_every=50):
        self.sink, self.flush_every, self.pending = sink, flush_every, []

    def add(self, tag, series, value, step):
        self.pending.append((tag, series, value.detach() if torch.is_tensor(value) else value, step))
        if len(self.pending) >= 7 * self.flush_every:
            self.flush()

    def flush(self):
        if not self.pending:
            return []
        vals = torch.stack([v.float() if torch.is_tensor(v) else torch.tensor(float(v)) for _, _, v, _ in
                            [(t, s, (x if not torch.is_tensor(x) else x.to(self._dev(), non_blocking=True)), k) for t, s, x, k in self.pending]])
        host = vals.cpu().tolist()                       # one device->host copy for everything pending
        out = [(t, s, h, k) for (t, s, _, k), h in zip(self.pending, host)]
        if self.sink is not None:
            for t, s, h, k in out:
                self.sink.add_scalars(t, {s: h}, k)
        self.pending = []
        return out

    def _dev(self):
        for _, _, v, _ in self.pending:
            if torch.is_tensor(v) and v.is_cuda:
                return v.device
        return torch.device("cpu")


def _scalars(self):
    d = getattr(self, "_deferred_scalars", None)
    if d is None:
        lg = getattr(self, "logger", None)
        d = DeferredScalars(getattr(lg, "experiment", None) if lg is not None else None)
        self._deferred_scalars = d
    return d


def training_step(self, batch, batch_idx):
    """step.py:14-25."""
    x, y = batch
    fl = getattr(self, "forward_and_loss", None)       # a model may fold `criterion(forward(x), y)` into one call (model/RetinaNet.py);
    result_dict = fl(x, y) if fl is not None else None  # None = it declined before running anything
    if result_dict is None:
        result_dict = self.criterion(self.forward(x), y)
    step = getattr(self, "global_step", batch_idx)
    for key, val in result_dict.items():
        _scalars(self).add("Loss/%s" % ("total" if key == "loss" else key), "Train", val, step)
    return result_dict


def _epoch_mean(self, outputs, series):
    for key in outputs[0].keys():
        avg = torch.stack([o[key].detach() for o in outputs]).mean()
        _scalars(self).add("Epoch/%s" % key, series, avg, getattr(self, "current_epoch", 0))
    return _scalars(self).flush()


def training_epoch_end(self, outputs):
    """step.py:29-41 (the per-parameter histograms and add_graph are TensorBoard features: left to the sink owner)."""
    return _epoch_mean(self, outputs, "Train")


def validation_step(self, batch, batch_idx):
    """step.py:43-50."""
    x, y = batch
    with torch.no_grad():
        out = self.forward(x)
        result_dict = self.criterion(out, y)
    if hasattr(self, "log_dict"):
        try:
            self.log_dict({"val_loss": result_dict["loss"]}, logger=True, on_epoch=True)
        except Exception:  # noqa: BLE001 - outside a PL trainer log_dict is unavailable
            pass
    return result_dict


def validation_epoch_end(self, outputs):
    """step.py:54-62."""
    out = _epoch_mean(self, outputs, "Val")
    if hasattr(self, "write_Best_model_path"):
        self.write_Best_model_path()
    return out


def test_step(self, batch, batch_idx):
    """step.py:64-100 without the cv2 overlays.  Keeps the reference's in-place `y[:, 2:] *= img_size` once per image
    in the batch (step.py:81; quirk 5 of SURVEY appendix B)."""
    self.inference = True
    x, y = batch
    with torch.no_grad():
        out = self.forward(x)
        suppress_output = self.non_max_suppression(out)
    for _ in range(x.shape[0]):
        y[:, 2:] *= self.img_size
    if self.checkname in ["RetinaNet", "SSD", "YOLOv5"]:
        return {"sample_metric": get_batch_statistics(suppress_output, y, iou_threshold=0.5), "label": y[:, 1].tolist()}
    return {"sample_metric": self.get_yolo_statistics(out, y, output_to_cpu=False), "detections": suppress_output}


def test_epoch_end(self, outputs):
    """step.py:102-150: AP per class (RetinaNet/SSD/YOLOv5) or per-grid statistics (YOLO v2/v3/v4)."""
    res = {}
    if self.checkname in ["RetinaNet", "SSD", "YOLOv5"]:
        metrics, labels = [], []
        for o in outputs:
            metrics += o["sample_metric"]
            labels += o["label"]
        if not metrics:
            return res
        tp, scores, pl = [np.concatenate(v, 0) for v in zip(*metrics)]
        precision, recall, AP, f1, ap_class = ap_per_class(tp, scores, pl, labels)
        res = {"precision": precision.mean(), "recall": recall.mean(), "AP": AP.mean(), "f1": f1.mean(),
               "ap_class": {int(c): AP[i] for i, c in enumerate(ap_class)}}
    else:
        for s in outputs[0]["sample_metric"].keys():
            cols = list(zip(*[o["sample_metric"][s] for o in outputs]))[:-1]
            names = ("cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj")
            res[s] = {n: float(np.array(c).mean()) for n, c in zip(names, cols)}
    sink = getattr(getattr(self, "logger", None), "experiment", None)
    if sink is not None:
        ep = getattr(self, "current_epoch", 0)
        for k, v in res.items():
            if isinstance(v, dict):
                for kk, vv in v.items():
                    sink.add_scalars(f"Test/{kk}/" if not isinstance(k, str) else f"Test/{k}", {f"grid_{k}" if not isinstance(k, str) else str(kk): vv}, ep)
            else:
                sink.add_scalars(f"Test/{k}", {"": v}, ep)
    return res
