"""Pins the oracle (oracle/yolo.py, oracle/boxes.py) to vectors captured from the reference itself."""
import numpy as np
import pytest
import torch

from oracle import boxes, yolo

V4_ANCHORS = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
V4_MASKS = [[0, 1, 2], [3, 4, 5], [6, 7, 8]]
CASES = ("plain", "dup", "tie", "oor", "oorb", "lab", "neg", "empty", "edge")
DENSE = ("iou_scores", "class_mask", "obj_mask", "noobj_mask", "tx", "ty", "tw", "th", "tcls", "tconf")


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("G", [13, 26])
@pytest.mark.parametrize("case", CASES)
def test_assign_bit_exact(golden, G, case):
    g = golden("yolo_assign")
    out = yolo.assign_targets(T(g[f"G{G}_pred_boxes"]), T(g[f"G{G}_pred_cls"]), T(g[f"G{G}_{case}_target"]),
                              T(g[f"G{G}_anchors"]), 0.5)
    for k in DENSE:
        ref = g[f"G{G}_{case}_{k}"]
        got = out[k].numpy()
        assert got.dtype == ref.dtype, k
        assert np.array_equal(got, ref), (k, int((got != ref).sum()))


def test_scaled_anchor_table_bits(golden):
    g = golden("yolo_assign")
    for G in (13, 26):
        assert np.array_equal(yolo.scaled_anchors(V4_ANCHORS[3:6], 416, G).numpy(), g[f"G{G}_anchors"])


def _close(a, b, rtol=1e-6, atol=1e-7):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.allclose(a, b, rtol=rtol, atol=atol, equal_nan=True)


@pytest.mark.parametrize("coord", ["smooth_l1_loss", "mse_loss"])
@pytest.mark.parametrize("cls", ["bce_loss", "ce_loss"])
@pytest.mark.parametrize("case", ["plain", "dup", "oor", "lab", "empty", "neg"])
def test_v4_loss_and_grad(golden, coord, cls, case):
    g = golden("yolo_loss")
    heads = [T(g[f"v4_head{i}"]).clone().requires_grad_() for i in range(3)]
    tag = f"v4_{coord}_{cls}_{case}"
    out = yolo.multiscale_region_loss_v4(heads, T(g[f"{tag}_target"]), V4_ANCHORS, V4_MASKS, 4, 416, coord=coord, cls=cls)
    out["loss"].backward()
    for k in yolo.KEYS:
        assert _close(out[k].detach().numpy(), g[f"{tag}_{k}"]), (k, out[k].item(), g[f"{tag}_{k}"])
    for i, h in enumerate(heads):
        assert _close(h.grad.numpy(), g[f"{tag}_grad{i}"], rtol=1e-5, atol=1e-9), i


def test_v4_saturated(golden):
    g = golden("yolo_loss")
    heads = [T(g[f"v4_sat_head{i}"]).clone().requires_grad_() for i in range(3)]
    out = yolo.multiscale_region_loss_v4(heads, T(g["v4_sat_target"]), V4_ANCHORS, V4_MASKS, 4, 416)
    out["loss"].backward()
    for k in yolo.KEYS:
        assert _close(out[k].detach().numpy(), g[f"v4_sat_{k}"]), k
    for i, h in enumerate(heads):
        assert _close(h.grad.numpy(), g[f"v4_sat_grad{i}"], rtol=1e-5, atol=1e-9)


def test_v3(golden):
    g = golden("yolo_loss")
    heads = [T(g[f"v3_head{i}"]).clone().requires_grad_() for i in range(3)]
    anchors = [[tuple(a) for a in s] for s in g["v3_anchors"].tolist()]
    out = yolo.multiscale_region_loss_v3(heads, T(g["v3_target"]), anchors, 4, 416, coord="mse_loss")
    out["loss"].backward()
    for k in yolo.KEYS:
        assert _close(out[k].detach().numpy(), g[f"v3_{k}"]), k
    for i, h in enumerate(heads):
        assert _close(h.grad.numpy(), g[f"v3_grad{i}"], rtol=1e-5, atol=1e-9)


@pytest.mark.parametrize("img", [288, 416])
def test_v2(golden, img):
    g = golden("yolo_loss")
    anchors = [(1.3221, 1.73145), (3.19275, 4.00944), (5.05587, 8.09892), (9.47112, 4.84053), (11.2364, 10.0071)]
    head = T(g[f"v2_{img}_head"]).clone().requires_grad_()
    out = yolo.region_loss_v2(head, T(g[f"v2_{img}_target"]), anchors, 20, img)
    out["loss"].backward()
    for k in yolo.KEYS:
        assert _close(out[k].detach().numpy(), g[f"v2_{img}_{k}"]), k
    assert _close(head.grad.numpy(), g[f"v2_{img}_grad"], rtol=1e-5, atol=1e-9)


def test_iou_family_bit_exact(golden):
    g = golden("iou")
    a, b, ax, bx = (T(g[k]) for k in ("a_xywh", "b_xywh", "a_xyxy", "b_xyxy"))
    assert np.array_equal(boxes.iou_plus1(ax, bx).numpy(), g["bbox_iou_xyxy"])
    assert np.array_equal(boxes.iou_plus1(a, b, x1y1x2y2=False).numpy(), g["bbox_iou_xywh"])
    assert np.array_equal(boxes.iou_plus1(ax[0:1], bx).numpy(), g["bbox_iou_bcast"])
    for mode in ("iou", "giou", "diou", "ciou"):
        assert np.array_equal(boxes.iou_v5(a.t(), b.t(), x1y1x2y2=False, mode=mode).numpy(), g[f"v5_{mode}_xywh"], equal_nan=True), mode
        assert np.array_equal(boxes.iou_v5(ax.t(), bx.t(), x1y1x2y2=True, mode=mode).numpy(), g[f"v5_{mode}_xyxy"], equal_nan=True), mode
    wh = torch.stack([boxes.wh_iou(an, T(g["wh_gwh"])) for an in T(g["wh_anchors"])])
    assert np.array_equal(wh.numpy(), g["wh_iou"])
    assert np.array_equal(boxes.iou_corner(T(g["corner_a"]), T(g["corner_b"])).numpy(), g["corner_iou"], equal_nan=True)
    assert np.array_equal(boxes.iou_matrix_plus1(a[:64], b[:16], "xywh").numpy(), g["retina_box_iou_xywh"])
    assert np.array_equal(boxes.iou_matrix_plus1(ax[:64], bx[:16], "xyxy").numpy(), g["retina_box_iou_xyxy"])


def test_yolo_statistics(golden):
    g = golden("yolo_stats")
    heads = [T(g["head0"]), T(g["head1"])]
    res = yolo.yolo_statistics(heads, T(g["target"]), V4_ANCHORS, [[3, 4, 5], [6, 7, 8]], 4, 416)
    for G in (26, 13):
        for i, name in enumerate(("cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj")):
            assert _close(res[G][i], g[f"G{G}_{name}"]), (G, name)
        assert _close(res[G][6].numpy(), g[f"G{G}_output"], rtol=1e-6, atol=1e-6)
