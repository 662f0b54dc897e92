"""Host logic of the loss-driven RetinaNet step that needs no GPU: the level-shape formula against the FPN itself, the decision to
decline, and `step.training_step`'s fallback."""
import argparse

import pytest
import torch

from objectdetectionpl_amd.LightningFunc import step as _step


def _model(C=3, size=64):
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    args = argparse.Namespace(model_name="RetinaNet", cls_criterion="focal_loss", coord_criterion="smooth_l1_loss", data_module="synthetic")
    RetinaNet.img_size = size
    try:
        return RetinaNet([str(i) for i in range(C)], args)
    finally:
        RetinaNet.img_size = 600


@pytest.mark.parametrize("hw", [(64, 96), (65, 130), (33, 47)])
def test_level_shapes_formula_matches_the_fpn_on_cpu(hw):
    model = _model().eval()
    x = torch.rand(1, 3, *hw)
    with torch.no_grad():
        feats = model.fpn(x)
    assert [(f.shape[2], f.shape[3]) for f in feats] == model._level_shapes(x)


def test_cpu_tensors_decline_before_anything_runs():
    model = _model().train()
    x = torch.rand(2, 3, 64, 64)
    y = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.3], [1, 2, 0.4, 0.6, 0.2, 0.5]])
    ran = []
    model.fpn.register_forward_hook(lambda *a: ran.append(1))
    assert model.loss_driven_mask(x, y) == 0
    assert model.forward_and_loss(x, y) is None and not ran


def test_training_step_falls_back_to_forward_plus_criterion_when_the_fold_declines():
    calls = []

    class M:
        global_step = 0

        def forward(self, x):
            calls.append("forward")
            return x * 2

        def criterion(self, out, y):
            calls.append("criterion")
            return {"loss": (out - y).sum()}

        def forward_and_loss(self, x, y):
            calls.append("fold")
            return None if self.decline else {"loss": torch.tensor(7.0)}

    m = M()
    m.decline = True
    d = _step.training_step(m, (torch.ones(2), torch.zeros(2)), 0)
    assert calls == ["fold", "forward", "criterion"] and float(d["loss"]) == 4.0
    del calls[:]
    m.decline = False
    d = _step.training_step(m, (torch.ones(2), torch.zeros(2)), 0)
    assert calls == ["fold"] and float(d["loss"]) == 7.0
