import numpy as np
import torch

from oracle import metrics


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_batch_statistics_exact(golden):
    g = golden("metrics")
    outs = [T(g[f"out{b}"]) if g[f"out{b}"].shape[0] else None for b in range(3)]
    st = metrics.batch_statistics(outs, T(g["targets"]), 0.5)
    assert len(st) == int(g["nstats"])
    for i, (tp, sc, lb) in enumerate(st):
        assert np.array_equal(tp, g[f"stat{i}_tp"]) and np.array_equal(sc, g[f"stat{i}_score"]) and np.array_equal(lb, g[f"stat{i}_label"])


def test_ap_per_class_exact(golden):
    g = golden("metrics")
    for pre in ("ap", "big"):
        p, r, ap, f1, uc = metrics.ap_per_class(g[f"{pre}_tp"], g[f"{pre}_conf"], g[f"{pre}_pred_cls"], g[f"{pre}_target_cls"])
        for name, v in (("p", p), ("r", r), ("ap", ap), ("f1", f1), ("classes", uc)):
            assert np.array_equal(v, g[f"{pre}_{name}"]), (pre, name)
    assert metrics.compute_ap(g["cap_recall"], g["cap_precision"]) == g["cap_ap"]
