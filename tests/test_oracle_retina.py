"""Pins oracle/retina.py (anchors, target encoding, softmax-focal + SmoothL1 loss) to the reference's vectors."""
import hashlib

import numpy as np
import pytest
import torch

from oracle import retina, yolo


def T(a):
    return torch.from_numpy(np.asarray(a))


def _close(a, b, rtol=1e-6, atol=1e-7):
    return np.allclose(np.asarray(a, np.float64), np.asarray(b, np.float64), rtol=rtol, atol=atol, equal_nan=True)


@pytest.mark.parametrize("wh", [(600, 600), (1333, 800), (128, 128), (96, 160)])
def test_anchor_boxes_bit_exact(golden, wh):
    g = golden("anchors")
    t = retina.anchor_boxes(torch.Tensor(list(wh)))
    tag = f"retina_{wh[0]}x{wh[1]}"
    assert t.shape[0] == int(g[tag + "_n"])
    assert np.array_equal(np.frombuffer(hashlib.sha256(t.numpy().tobytes()).digest(), dtype=np.uint8), g[tag + "_sha256"])
    assert np.array_equal(t[:64].numpy(), g[tag + "_head"]) and np.array_equal(t[-64:].numpy(), g[tag + "_tail"])
    assert np.array_equal(retina.anchor_wh_table().numpy(), g["retina_wh_table"])


def test_yolo_scaled_anchor_tables(golden):
    g = golden("anchors")
    A9 = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
    v2 = [(1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071)]
    for name, anc in (("v4s0", A9[0:3]), ("v4s1", A9[3:6]), ("v4s2", A9[6:9]), ("v2", v2)):
        for G in (9, 13, 26, 52):
            assert np.array_equal(yolo.scaled_anchors(anc, 416, G).numpy(), g[f"yolo_{name}_G{G}"])


@pytest.mark.parametrize("coord", ["smooth_l1_loss", "mse_loss"])
def test_retina_loss_and_grads(golden, coord):
    g = golden("retina")
    loc = T(g["r128_loc"]).clone().requires_grad_()
    cls = T(g["r128_cls"]).clone().requires_grad_()
    out = retina.retina_loss(loc, cls, T(g["r128_target"]), T(g["r128_anchors"]), 4, 128, coord=coord)
    out["loss"].backward()
    tag = f"r128_{coord}"
    for k in ("loss", "Localization", "Classification"):
        assert _close(out[k].detach().numpy(), g[f"{tag}_{k}"]), k
    assert _close(loc.grad.numpy(), g[f"{tag}_grad_loc"], rtol=1e-5, atol=1e-9)
    assert _close(cls.grad.numpy(), g[f"{tag}_grad_cls"], rtol=1e-5, atol=1e-9)


def test_retina_encoding_bit_exact(golden):
    g = golden("retina")
    anchors, tgt = T(g["r128_anchors"]), T(g["r128_target"])
    from oracle.boxes import iou_matrix_plus1
    for bid in range(2):
        boxes = tgt[tgt[:, 0] == bid][:, 2:] * 128
        mi, mid = iou_matrix_plus1(anchors, boxes, "xywh").max(1)
        assert np.array_equal(mi.numpy(), g[f"r128_maxiou_{bid}"]) and np.array_equal(mid.numpy(), g[f"r128_maxid_{bid}"])


def test_retina_saturated(golden):
    g = golden("retina")
    loc = T(g["rsat_loc"]).clone().requires_grad_()
    cls = T(g["rsat_cls"]).clone().requires_grad_()
    out = retina.retina_loss(loc, cls, T(g["rsat_target"]), T(g["rsat_anchors"]), 4, 128)
    out["loss"].backward()
    for k in ("loss", "Localization", "Classification"):
        assert _close(out[k].detach().numpy(), g[f"rsat_{k}"]), k
    assert _close(cls.grad.numpy(), g["rsat_grad_cls"], rtol=1e-5, atol=1e-9)
    assert _close(loc.grad.numpy(), g["rsat_grad_loc"], rtol=1e-5, atol=1e-9)


def test_softmax_focal(golden):
    g = golden("retina")
    x = T(g["focal_x"]).clone().requires_grad_()
    l = retina.softmax_focal(x, T(g["focal_y"]), 6, "sum")
    l.backward()
    assert _close(l.item(), g["focal_sum"]) and _close(x.grad.numpy(), g["focal_grad"], rtol=1e-5, atol=1e-9)
    assert np.array_equal(retina.softmax_focal(x.detach(), T(g["focal_y"]), 6, "none").numpy(), g["focal_none"])
