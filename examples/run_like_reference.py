#!/usr/bin/env python3
"""The reference's `run.py:26-118`, call for call, on this package: the yaml is flattened into one Namespace, a data module and
a model are built from it, `read_Best_model_path`, TensorBoardLogger, ModelCheckpoint / EarlyStopping / LearningRateMonitor /
GPUStatsMonitor, `Trainer.from_argparse_args(config, callbacks=[...], logger=logger)`, optional `.tune`, `.fit`, `.test`.
The `pytorch_lightning` names resolve to a real PL when one is importable, else to the in-repo stand-in
(`objectdetectionpl_amd/compat`).  Differences from run.py, all forced by what exists here: the data module is the synthetic
one (the reference's datasets are Windows paths) and `img_size` may come from the yaml (run.py:35-38
forces 416 for YOLOv2; BASELINE config 1 asks for 288).

    python examples/run_like_reference.py [examples/configs/config.yaml]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from objectdetectionpl_amd import compat

compat.install()                                   # no-op when pytorch_lightning is installed

import yaml                                                                             # noqa: E402
from pytorch_lightning import Trainer                                                   # noqa: E402
from pytorch_lightning.callbacks import GPUStatsMonitor, LearningRateMonitor, ModelCheckpoint   # noqa: E402
from pytorch_lightning.callbacks.early_stopping import EarlyStopping                    # noqa: E402
from pytorch_lightning.loggers import TensorBoardLogger                                 # noqa: E402

from objectdetectionpl_amd.dataset.Synthetic import SyntheticCocoModule                  # noqa: E402
from objectdetectionpl_amd.model.RetinaNet import RetinaNet                              # noqa: E402
from objectdetectionpl_amd.model.SSD import SSD                                          # noqa: E402
from objectdetectionpl_amd.model.YOLOV2 import YOLOv2                                    # noqa: E402
from objectdetectionpl_amd.model.YOLOV3 import YOLOv3                                    # noqa: E402
from objectdetectionpl_amd.model.YOLOV4 import YOLOv4                                    # noqa: E402
from objectdetectionpl_amd.model.YOLOV5 import YOLOv5                                    # noqa: E402


def load_config(args):
    with open(args.config) as f:
        config = yaml.load(f, Loader=yaml.SafeLoader)
    for key, value in config.items():
        if isinstance(value, dict):
            for inside_key, inside_value in value.items():
                setattr(args, inside_key, inside_value)
        else:
            setattr(args, key, value)
    if not getattr(args, "img_size", 0):
        args.img_size = {"RetinaNet": 600, "SSD": 300, "YOLOv5": 640}.get(args.model_name, 416)
    return args


def load_data(args):
    dm = SyntheticCocoModule(batch_size=args.batch_size, img_size=args.img_size, num_classes=getattr(args, "num_classes", 20),
                             steps_per_epoch=getattr(args, "steps_per_epoch", 8),
                             device="cuda:%d" % int(os.environ.get("LOCAL_RANK", "0")))
    dm.setup(args.stage)
    return dm


def load_model(args, dm):
    cls = {"RetinaNet": RetinaNet, "SSD": SSD, "YOLOv2": YOLOv2, "YOLOv3": YOLOv3, "YOLOv4": YOLOv4, "YOLOv5": YOLOv5}[args.model_name]
    model = cls(dm.get_class(), args)
    if args.model_name == "YOLOv2":
        model.criterion.img_dim = model.img_size = args.img_size                     # device-agnostic classes accept an img_size override
    return model


def main(argv=None, out_dir="log_dir"):
    parser = argparse.ArgumentParser()
    parser.add_argument("config", nargs="?", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "configs", "config.yaml"),
                        help="YAML configuration file")
    args = parser.parse_args(argv)

    config = load_config(args)
    dm = load_data(args)
    model = load_model(args, dm)
    model.dir = os.path.join(out_dir, dm.name, model.checkname)

    model.read_Best_model_path()

    root_dir = os.path.join(out_dir, dm.name)
    logger = TensorBoardLogger(root_dir, name=model.checkname, default_hp_metric=False)

    checkpoint_callback = ModelCheckpoint(monitor="val_loss", dirpath=os.path.join(root_dir, model.checkname),
                                          filename=model.checkname + "-{epoch:02d}-{val_loss:.2f}", save_top_k=3, mode="min", verbose=True)
    setattr(model, "checkpoint_callback", checkpoint_callback)
    early_stop_callback = EarlyStopping(monitor="val_loss", min_delta=0.00, patience=3, verbose=False, mode="min")
    lr_monitor = LearningRateMonitor(logging_interval="epoch")
    gpu_stats = GPUStatsMonitor()

    trainer = Trainer.from_argparse_args(config, callbacks=[checkpoint_callback, early_stop_callback, lr_monitor, gpu_stats], logger=logger)

    if args.tune:
        trainer.tune(model, datamodule=dm)
    trainer.fit(model, datamodule=dm)

    dm.setup("test")
    result = trainer.test(model, datamodule=dm)
    return dict(trainer=trainer, model=model, logger=logger, checkpoint=checkpoint_callback, early_stop=early_stop_callback,
                lr_monitor=lr_monitor, gpu_stats=gpu_stats, test=result)


if __name__ == "__main__":
    r = main()
    print("best checkpoint:", r["checkpoint"].best_model_path)
    print("test:", r["test"])
