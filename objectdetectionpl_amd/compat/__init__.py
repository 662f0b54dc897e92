"""Stand-ins for third-party packages the reference imports and this image cannot install.

`compat/pytorch_lightning` exposes, under PL's own module paths, the API subset `run.py:1-4,86-118`, `LightningFunc/step.py`
and `LightningFunc/lightningUtils.py` use.  Put this directory on `sys.path` (or call `install()`) and
`from pytorch_lightning import Trainer`, `from pytorch_lightning.loggers import TensorBoardLogger`,
`from pytorch_lightning.callbacks import ModelCheckpoint, LearningRateMonitor, GPUStatsMonitor`,
`from pytorch_lightning.callbacks.early_stopping import EarlyStopping` resolve here.  A real pytorch_lightning, when
importable, always wins: install() does nothing then."""
import importlib
import os
import sys


def install(force=False):
    """Make `import pytorch_lightning` resolve to the in-repo stand-in unless the real package is importable."""
    if not force:
        try:
            m = importlib.import_module("pytorch_lightning")
            return m
        except ImportError:
            pass
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    for k in [k for k in sys.modules if k == "pytorch_lightning" or k.startswith("pytorch_lightning.")]:
        del sys.modules[k]
    return importlib.import_module("pytorch_lightning")
