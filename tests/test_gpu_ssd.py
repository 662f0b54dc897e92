"""GPU parity: SSD loss (C ABI) vs the reference's golden vectors."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("cls", ["ce_loss", "focal_loss"])
@pytest.mark.parametrize("coord", ["smooth_l1_loss", "mse_loss"])
@pytest.mark.parametrize("tname", ["t1", "t2"])
def test_ssd_loss_vs_golden(golden, cls, coord, tname):
    import torch.nn as nn
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss, focal_loss
    g = golden("ssd")
    d = torch.device("cuda:0")
    o = T(g["off"]).to(d).requires_grad_(); c = T(g["cls"]).to(d).requires_grad_()
    tgt = T(g["target" if tname == "t1" else "target2"]).to(d)
    crit = SSDLoss(T(g["priors"]), focal_loss if cls == "focal_loss" else nn.CrossEntropyLoss, coord, 6, 300)
    out = crit((o, c), tgt)
    out["loss"].backward()
    tag = f"{cls}_{coord}_{tname}"
    for k in ("loss", "Localization", "Classification"):
        a, b = out[k].item(), float(np.asarray(g[f"{tag}_{k}"]).reshape(-1)[0])
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (k, a, b)
    for got, name in ((o.grad, "grad_off"), (c.grad, "grad_cls")):
        ref = g[f"{tag}_{name}"]; got = got.cpu().numpy()
        assert np.array_equal(got != 0, ref != 0), name                      # matched set / hard negatives: index work
        assert np.allclose(got, ref, rtol=1e-4, atol=1e-6 * np.abs(ref).max() + 1e-12), (name, np.abs(got - ref).max())
    assert int(crit.last_counts[0]) == (3 if tname == "t1" else 2)
