import numpy as np
import pytest
import torch

from oracle import yolov5

V5_ANCHORS = [[10, 13, 16, 30, 33, 23], [30, 61, 62, 45, 59, 119], [116, 90, 156, 198, 373, 326]]


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("case", ["plain", "dup", "empty"])
def test_assign_rows_bit_exact(golden, case):
    g = golden("yolov5")
    anc = yolov5.grid_anchors(V5_ANCHORS)
    assert np.array_equal(anc.numpy(), g["grid_anchors"])
    for i, G in enumerate((16, 8, 4)):
        b, a, gj, gi, tbox, an, tcls = yolov5.assign_layer(G, T(g[f"{case}_target"]), anc[i])
        for name, v in (("b", b), ("a", a), ("gj", gj), ("gi", gi), ("tbox", tbox), ("anch", an), ("tcls", tcls)):
            assert np.array_equal(v.numpy(), g[f"{case}_L{i}_{name}"]), (i, name)


@pytest.mark.parametrize("case", ["plain", "dup", "empty"])
def test_v5_loss_and_grads(golden, case):
    g = golden("yolov5")
    heads = [T(g[f"{case}_head{i}"]).clone().requires_grad_() for i in range(3)]
    out = yolov5.v5_loss(heads, T(g[f"{case}_target"]), V5_ANCHORS, 4)
    out["loss"].backward()
    for k in ("loss", "Localization", "Classification", "Conf_obj"):
        assert np.allclose(out[k].detach().numpy(), g[f"{case}_{k}"], rtol=1e-6, atol=1e-7), k
    for i, h in enumerate(heads):
        assert np.allclose(h.grad.numpy(), g[f"{case}_grad{i}"], rtol=1e-5, atol=1e-9), i
