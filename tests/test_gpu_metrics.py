import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def test_batch_statistics_vs_golden(golden):
    from objectdetectionpl_amd.LightningFunc.accuracy import get_batch_statistics
    g = golden("metrics")
    outs = [T(g[f"out{b}"]) if g[f"out{b}"].shape[0] else None for b in range(3)]
    st = get_batch_statistics(outs, T(g["targets"]), 0.5)
    assert len(st) == int(g["nstats"])
    for i, (tp, sc, lb) in enumerate(st):
        assert tp.dtype == np.float64 and np.array_equal(tp, g[f"stat{i}_tp"])
        assert np.array_equal(sc, g[f"stat{i}_score"]) and np.array_equal(lb, g[f"stat{i}_label"])


@pytest.mark.parametrize("pre", ["ap", "big"])
def test_ap_per_class_vs_golden(golden, pre):
    from objectdetectionpl_amd.LightningFunc.accuracy import ap_per_class
    g = golden("metrics")
    p, r, ap, f1, uc = ap_per_class(g[f"{pre}_tp"], g[f"{pre}_conf"], g[f"{pre}_pred_cls"], g[f"{pre}_target_cls"])
    assert np.array_equal(uc, g[f"{pre}_classes"]) and uc.dtype == np.int32
    assert np.array_equal(p, g[f"{pre}_p"]) and np.array_equal(r, g[f"{pre}_r"])          # integer counts / exact divisions
    assert np.allclose(ap, g[f"{pre}_ap"], rtol=1e-12, atol=1e-15) and np.allclose(f1, g[f"{pre}_f1"], rtol=1e-12, atol=1e-15)


def test_get_yolo_statistics_vs_golden(golden):
    from objectdetectionpl_amd.LightningFunc.accuracy import get_yolo_statistics
    g = golden("yolo_stats")

    class M:
        anchors = [[12, 16], [19, 36], [40, 28], [36, 75], [76, 55], [72, 146], [142, 110], [192, 243], [459, 401]]
        anch_masks = [[3, 4, 5], [6, 7, 8]]
        num_classes, img_size, ignore_thres = 4, 416, 0.5
    res = get_yolo_statistics(M(), [T(g["head0"]), T(g["head1"])], T(g["target"]))
    for G in (26, 13):
        for i, name in enumerate(("cls_acc", "recall50", "recall75", "precision", "conf_obj", "conf_noobj")):
            a, b = float(res[G][i]), float(g[f"G{G}_{name}"])
            assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (G, name, a, b)
        assert np.allclose(res[G][6].numpy(), g[f"G{G}_output"], rtol=1e-5, atol=1e-5)
