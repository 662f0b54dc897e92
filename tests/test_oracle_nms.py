"""Pins oracle/nms.py and oracle/retina.greedy_nms to the reference's own NMS outputs."""
import numpy as np
import pytest
import torch

from oracle import nms, retina


def T(a):
    return torch.from_numpy(np.asarray(a))


@pytest.mark.parametrize("name,nh", [("a", 3), ("b", 2), ("c", 1)])
def test_yolo_weighted_nms(golden, name, nh):
    g = golden("nms")
    heads = [T(g[f"yolo_{name}_head{i}"]) for i in range(nh)]
    out = nms.yolo_weighted_nms(heads)
    for b, o in enumerate(out):
        if bool(g[f"yolo_{name}_none{b}"]):
            assert o is None
        else:
            assert np.array_equal(o.numpy(), g[f"yolo_{name}_out{b}"]), (name, b)


def test_v5_reinterpretation(golden):
    g = golden("nms")
    v5 = [T(g["v5_head0"]), T(g["v5_head1"])]
    flat = [h.reshape(h.shape[0], 3 * h.shape[4], h.shape[2], h.shape[3]) for h in v5]    # same memory, NCHW reading
    out = nms.yolo_weighted_nms(flat)
    for b, o in enumerate(out):
        assert np.array_equal(o.numpy(), g[f"v5_out{b}"])


@pytest.mark.parametrize("name", ["typ", "few", "none", "many"])
def test_retina_greedy_nms(golden, name):
    g = golden("nms")
    out = retina.greedy_nms(T(g[f"ret_{name}_loc"]), T(g[f"ret_{name}_cls"]), T(g["ret_anchors"]))
    for b, o in enumerate(out):
        ref = g[f"ret_{name}_out{b}"]
        assert o.shape == ref.shape and np.array_equal(o.numpy(), ref), (name, b, o.shape, ref.shape)
