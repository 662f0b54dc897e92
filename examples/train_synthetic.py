#!/usr/bin/env python3
"""The reference's `run.py:69-118` flow on synthetic data, without pytorch_lightning: pick a model class by name, attach a
ModelCheckpoint, fit, test.  Single GPU:  python examples/train_synthetic.py --model YOLOv4 --epochs 2
Data parallel:  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 examples/train_synthetic.py --model YOLOv4"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd.dataset.Synthetic import SyntheticCocoModule
from objectdetectionpl_amd.LightningFunc.lightningUtils import ModelCheckpoint
from objectdetectionpl_amd.trainer import Trainer

MODELS = {"YOLOv2": ("YOLOV2", 416), "YOLOv3": ("YOLOV3", 416), "YOLOv4": ("YOLOV4", 416), "YOLOv5": ("YOLOV5", 640),
          "RetinaNet": ("RetinaNet", 600), "SSD": ("SSD", 300)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="YOLOv4", choices=sorted(MODELS))
    ap.add_argument("--epochs", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--steps-per-epoch", type=int, default=20)
    ap.add_argument("--classes", type=int, default=20)
    ap.add_argument("--raw-frames", action="store_true", help="synthetic uint8 frames through the device collate (resize/flips/LUTs)")
    ap.add_argument("--out", default="log_dir")
    a = ap.parse_args()
    mod, img = MODELS[a.model]
    args = argparse.Namespace(model_name=a.model, type="Yolov5s", data_module="synthetic", optimizer="Adam", lr=1e-3, lr_decay=0,
                              lr_scheduler="ReduceLROnPlateau", momentum=0.9, weight_decay=1e-5, alpha=0.95,
                              cls_criterion="focal_loss" if a.model == "RetinaNet" else ("ce_loss" if a.model == "SSD" else "bce_loss"),
                              coord_criterion="smooth_l1_loss")
    dev = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
    dm = SyntheticCocoModule(a.batch, img, a.classes, steps_per_epoch=a.steps_per_epoch, device=dev, channels_last=True,
                             raw_frames=a.raw_frames, seed=1)          # rank r draws from seed 1 + r
    model = getattr(__import__(f"objectdetectionpl_amd.model.{mod}", fromlist=[a.model]), a.model)(dm.get_class(), args)
    model = model.to(dev).to(memory_format=torch.channels_last)
    ckpt = ModelCheckpoint(monitor="val_loss", dirpath=os.path.join(a.out, dm.name, model.checkname),
                           filename=model.checkname + "-{epoch:02d}-{val_loss:.2f}", save_top_k=3, mode="min", verbose=True)
    trainer = Trainer(max_epochs=a.epochs, limit_val_batches=4, limit_test_batches=4, callbacks=[ckpt], device=dev)
    for kind, epoch, scalars in trainer.fit(model, dm):
        print(kind, epoch, {t: round(v, 4) for t, _, v, _ in scalars if t.startswith("Epoch/")})
    print("best checkpoint:", ckpt.best_model_path)
    print("test:", trainer.test(model, dm))


if __name__ == "__main__":
    main()
