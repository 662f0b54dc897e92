import numpy as np
import pytest
import torch

from oracle import ssd


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_priors_and_match_bit_exact(golden):
    g = golden("ssd")
    pri = ssd.dboxes()
    assert np.array_equal(pri.numpy(), g["priors"])
    ann, mb = ssd.match(pri, T(g["target"])[:3, 2:])
    assert np.array_equal(ann.numpy(), g["match_ann"]) and np.array_equal(mb.numpy().astype(bool), g["match_bin"].astype(bool))


@pytest.mark.parametrize("cls", ["ce_loss", "focal_loss"])
@pytest.mark.parametrize("coord", ["smooth_l1_loss", "mse_loss"])
@pytest.mark.parametrize("tname", ["t1", "t2"])
def test_ssd_loss_and_grads(golden, cls, coord, tname):
    g = golden("ssd")
    o = T(g["off"]).clone().requires_grad_(); c = T(g["cls"]).clone().requires_grad_()
    tgt = T(g["target" if tname == "t1" else "target2"])
    out = ssd.ssd_loss(o, c, tgt, T(g["priors"]), cls, coord)
    out["loss"].backward()
    tag = f"{cls}_{coord}_{tname}"
    for k in ("loss", "Localization", "Classification"):
        assert np.allclose(out[k].detach().numpy(), g[f"{tag}_{k}"], rtol=1e-6, atol=1e-7), k
    assert np.allclose(o.grad.numpy(), g[f"{tag}_grad_off"], rtol=1e-5, atol=1e-9)
    assert np.allclose(c.grad.numpy(), g[f"{tag}_grad_cls"], rtol=1e-5, atol=1e-9)
