"""`write_Best_model_path` / `read_Best_model_path` of the reference (`LightningFunc/lightningUtils.py:17-37`) and a
minimal `ModelCheckpoint` with the constructor the reference uses (`run.py:86-93`: monitor / dirpath / filename /
save_top_k / mode), for machines without pytorch_lightning.  Checkpoints are `{"state_dict", "epoch", monitor}` files
(PL's key for the weights), so a PL checkpoint's `state_dict` and ours load the same way."""
import os

import torch


class ModelCheckpoint:
    def __init__(self, monitor="val_loss", dirpath=".", filename="model-{epoch:02d}-{val_loss:.2f}", save_top_k=3, mode="min", verbose=False):
        self.monitor, self.dirpath, self.filename, self.save_top_k, self.mode, self.verbose = monitor, dirpath, filename, save_top_k, mode, verbose
        self.best_model_path, self.best_model_score, self._kept = "", None, []

    def _better(self, a, b):
        return a < b if self.mode == "min" else a > b

    def on_validation_end(self, model, epoch, value):
        """Keep the `save_top_k` best epochs; returns the path written or None."""
        if value != value:                                                   # NaN never wins
            return None
        if len(self._kept) >= self.save_top_k > 0 and not self._better(value, self._kept[-1][0]):
            return None
        os.makedirs(self.dirpath, exist_ok=True)
        name = self.filename.replace("{epoch:02d}", "epoch=%02d" % epoch).replace("{%s:.2f}" % self.monitor, "%s=%.2f" % (self.monitor, value))
        path = os.path.join(self.dirpath, name + ".ckpt")
        torch.save({"state_dict": model.state_dict(), "epoch": epoch, self.monitor: value}, path)
        self._kept.append((value, path))
        self._kept.sort(key=lambda t: t[0], reverse=self.mode != "min")
        while self.save_top_k > 0 and len(self._kept) > self.save_top_k:
            _, old = self._kept.pop()
            if os.path.exists(old):
                os.remove(old)
        self.best_model_score, self.best_model_path = self._kept[0]
        if self.verbose:
            print("checkpoint: %s" % path)
        return path


def write_Best_model_path(self):
    """lightningUtils.py:17-24."""
    cb = getattr(self, "checkpoint_callback", None)
    best = getattr(cb, "best_model_path", "")
    if best != "":
        os.makedirs(self.dir, exist_ok=True)
        with open(os.path.join(self.dir, "best_model_path.txt"), "w", encoding="utf-8") as f:
            f.write(best + "\n")


def read_Best_model_path(self):
    """lightningUtils.py:26-37 -- with the one deliberate difference that the weights ARE loaded into `self` (the reference
    calls the classmethod `load_from_checkpoint` and discards the returned model: SURVEY appendix B, quirk 9).
    Returns the path loaded, or None."""
    fn = os.path.join(self.dir, "best_model_path.txt")
    if not os.path.exists(fn):
        return None
    with open(fn, "r", encoding="utf-8") as f:
        path = f.readline().strip()
    if not path or not os.path.exists(path):
        return None
    ck = torch.load(path, map_location="cpu", weights_only=True)
    self.load_state_dict(ck["state_dict"])
    return path
