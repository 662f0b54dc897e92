"""`write_Best_model_path` / `read_Best_model_path` of the reference (`LightningFunc/lightningUtils.py:17-37`).  The
`ModelCheckpoint` they read `best_model_path` from lives in `objectdetectionpl_amd/callbacks.py` (run.py:88-95)."""
import os

import torch


from ..callbacks import ModelCheckpoint  # noqa: E402,F401  (kept importable from here: round-1 location)


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
