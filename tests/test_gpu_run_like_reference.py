"""run.py's call sequence (run.py:70-118) through the `pytorch_lightning` names, end to end on the device at
BASELINE config 1's shape (YOLOv2, 288x288, batch 2; the yaml's `yaml_test` block makes it 2 epochs x 5 batches exactly as the
reference's shipped config does)."""
import importlib.util
import json
import os

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_run_py_sequence_on_config1_shape(tmp_path):
    spec = importlib.util.spec_from_file_location("run_like_reference", os.path.join(REPO, "examples", "run_like_reference.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    r = mod.main([os.path.join(REPO, "examples", "configs", "config.yaml")], out_dir=str(tmp_path))
    tr, model = r["trainer"], r["model"]
    assert type(tr).__name__ == "Trainer" and tr.max_epochs == 2 and tr.accumulate == 8 and tr.limits == {"train": 5, "val": 5, "test": 5}
    assert tr.current_epoch == 1 and tr.global_step == 10                       # 2 epochs x 5 batches (yaml_test wins over training.max_epochs)
    assert model.img_size == 288 and model.criterion.grid_size == 9             # config 1: 288 / 32
    # ModelCheckpoint wrote, best path recorded by validation_epoch_end -> write_Best_model_path (step.py:62)
    ck = r["checkpoint"]
    assert ck.best_model_path and os.path.exists(ck.best_model_path)
    assert open(os.path.join(model.dir, "best_model_path.txt")).read().strip() == ck.best_model_path
    assert r["early_stop"].wait_count <= 1 and not tr.should_stop
    assert len(r["lr_monitor"].lrs["lr-Adam"]) == 2 and len(r["gpu_stats"].stats) == 2
    # the scalar sink got the seven training-loss keys of step.py:21-23 (deferred, flushed at epoch end) and val_loss
    lines = [json.loads(l) for l in open(os.path.join(r["logger"].log_dir, "scalars.jsonl"))]
    tags = {l["tag"] for l in lines}
    assert {"Loss/total", "Loss/Localization", "Loss/Size", "Loss/Conf", "Loss/Classification", "Loss/Conf_obj", "Loss/Conf_noobj",
            "val_loss", "lr-Adam"} <= tags
    assert sum(1 for l in lines if l["tag"] == "Loss/total" and l.get("series") == "Train") == 10
    # test: per-grid statistics of get_yolo_statistics (step.py:102-150, YOLO branch)
    assert r["test"] and all(set(v) == {"cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj"} for v in r["test"].values())
