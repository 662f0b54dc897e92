"""GPU parity: batched NMS (C ABI) vs the reference's own NMS outputs (golden) and the oracle."""
import numpy as np
import pytest
import torch

from oracle import nms as oracle_nms
from oracle import retina as oracle_retina

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def _check_yolo(out, g, prefix, nimg):
    for b in range(nimg):
        ref = g[f"{prefix}_out{b}"]
        if f"{prefix}_none{b}" in g.files and bool(g[f"{prefix}_none{b}"]):
            assert out[b] is None
            continue
        got = out[b].cpu().numpy()
        assert got.shape == ref.shape, (prefix, b, got.shape, ref.shape)
        # kept set / order / labels / confidences are index work: exact; merged boxes are fp32 sums
        assert np.array_equal(got[:, 4:], ref[:, 4:]), (prefix, b)
        assert np.allclose(got[:, :4], ref[:, :4], rtol=1e-5, atol=1e-4), np.abs(got[:, :4] - ref[:, :4]).max()


@pytest.mark.parametrize("name,nh", [("a", 3), ("b", 2), ("c", 1)])
def test_yolo_nms_vs_golden(golden, name, nh):
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    g = golden("nms")
    out = yolo_non_max_suppression([T(g[f"yolo_{name}_head{i}"]) for i in range(nh)])
    _check_yolo(out, g, f"yolo_{name}", 2)


def test_yolo_nms_v5_layout(golden):
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    g = golden("nms")
    out = yolo_non_max_suppression([T(g["v5_head0"]), T(g["v5_head1"])])
    _check_yolo(out, g, "v5", 2)


@pytest.mark.parametrize("name", ["typ", "few", "none", "many"])
def test_greedy_nms_vs_golden(golden, name):
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    g = golden("nms")
    out = greedy_non_max_suppression((T(g[f"ret_{name}_loc"]), T(g[f"ret_{name}_cls"])), T(g["ret_anchors"]))
    for b, o in enumerate(out):
        ref = g[f"ret_{name}_out{b}"]
        got = o.cpu().numpy()
        assert got.shape == ref.shape, (name, b, got.shape, ref.shape)
        assert np.array_equal(got[:, 6], ref[:, 6]) and np.array_equal(got[:, 4], ref[:, 4])
        assert np.allclose(got[:, :4], ref[:, :4], rtol=1e-5, atol=1e-4)
        assert np.allclose(got[:, 5], ref[:, 5], rtol=1e-6, atol=1e-7)


def test_greedy_nms_single_candidate_raises_like_the_reference():
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    an = torch.tensor([[10., 10., 8., 8.], [30., 30., 8., 8.], [50., 50., 8., 8.]]).cuda()
    loc = torch.zeros(1, 3, 4).cuda()
    cls = torch.full((1, 3, 2), -9.0).cuda(); cls[0, 1, 0] = 3.0
    with pytest.raises(IndexError):
        greedy_non_max_suppression((loc, cls), an)


def test_yolo_nms_config2_size_vs_oracle():
    """BASELINE config-2 candidate count (10 647 / image) with clustered valid boxes: kept set identical to the oracle."""
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    g = torch.Generator().manual_seed(61)
    B, C = 2, 80
    heads = []
    for G in (52, 26, 13):
        h = torch.randn(B, 3, 5 + C, G, G, generator=g)
        centres = torch.rand(B, 3, 2, 1, 1, generator=g) * 300 + 50
        h[:, :, 0:2] = centres + torch.randn(B, 3, 2, G, G, generator=g) * 40
        h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 80 + 20
        h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.8 - 0.6
        h[:, :, 5:] = torch.rand(B, 3, C, G, G, generator=g)
        heads.append(h.view(B, 3 * (5 + C), G, G).contiguous())
    ref = oracle_nms.yolo_weighted_nms(heads)
    for ow in (False, True):
        out = yolo_non_max_suppression([h.cuda() for h in heads], one_workgroup=ow)
        for r, o in zip(ref, out):
            o = o.cpu()
            assert o.shape == r.shape
            assert torch.equal(o[:, 4:], r[:, 4:])
            assert torch.allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)


def _clustered_heads(g, B, C, grids, conf_shift, spread=40.0):
    heads = []
    for G in grids:
        h = torch.randn(B, 3, 5 + C, G, G, generator=g)
        centres = torch.rand(B, 3, 2, 1, 1, generator=g) * 300 + 50
        h[:, :, 0:2] = centres + torch.randn(B, 3, 2, G, G, generator=g) * spread
        h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 80 + 20
        h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.8 + conf_shift
        h[:, :, 5:] = torch.rand(B, 3, C, G, G, generator=g)
        heads.append(h.view(B, 3 * (5 + C), G, G).contiguous())
    return heads


def _same_as_oracle(heads, **kw):
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    ref = oracle_nms.yolo_weighted_nms(heads, **kw)
    outs = [yolo_non_max_suppression([h.cuda() for h in heads], one_workgroup=ow, **kw) for ow in (False, True)]
    for out in outs:
        for r, o in zip(ref, out):
            if r is None:
                assert o is None
                continue
            o = o.cpu()
            assert o.shape == r.shape
            assert torch.equal(o[:, 4:], r[:, 4:])
            assert torch.allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)
    for a, b in zip(*outs):                  # class-split pipeline (>= 8 classes) vs one workgroup per image
        if a is None or b is None:
            assert a is None and b is None
        else:                                # same kept set, order, labels and confidences; the merged boxes are the same sums in another
            assert torch.equal(a[:, 4:], b[:, 4:]) and torch.allclose(a[:, :4], b[:, :4], rtol=1e-5, atol=1e-3)     # association order


def test_yolo_nms_class_split_pipeline_at_config2_size_and_with_a_dominant_class():
    """[r4] bucket -> one workgroup per (image, class) -> rank order of the kept rows: 80 classes at 10 647 candidates / image (every
    segment in one wave's registers); 12 classes of which one holds ~70 % of ~3 000 valid candidates (a long segment: rank by counting
    + workgroup sweeps) next to classes with none; both against the oracle and against the one-workgroup kernel."""
    g = torch.Generator().manual_seed(71)
    _same_as_oracle(_clustered_heads(g, 3, 80, (52, 26, 13), conf_shift=-0.6))
    heads = _clustered_heads(g, 2, 12, (26, 13), conf_shift=0.8, spread=30.0)
    for h, G in zip(heads, (26, 13)):
        v = h.view(2, 3, 17, G, G)
        dom = torch.rand(2, 3, G, G, generator=g) < 0.7
        v[:, :, 5 + 4][dom] = 2.0                                   # class 4 wins on ~70 % of the cells
        v[:, :, 5 + 9] = -1.0                                       # class 9 never wins
    _same_as_oracle(heads)


def test_yolo_nms_long_class_lists():
    """Two classes and ~1500 valid candidates per image: per-class lists far above the 256 register-resident entries,
    so the LDS/L2 walk of the class-parallel kernel runs (and many merges per pivot)."""
    g = torch.Generator().manual_seed(67)
    _same_as_oracle(_clustered_heads(g, 2, 2, (26, 13), conf_shift=0.3, spread=25.0))


def test_yolo_nms_single_class_and_sparse():
    g = torch.Generator().manual_seed(68)
    _same_as_oracle(_clustered_heads(g, 2, 1, (13,), conf_shift=-0.5))
    heads = _clustered_heads(g, 3, 20, (13, 7), conf_shift=-3.5)       # a handful of valid candidates, or none, per image
    heads[0][1, :, :] = heads[0][1, :, :].clone()
    heads[0].view(3, 3, 25, 13, 13)[1, :, 4] = -9.0                      # image 1: nothing valid on this scale ...
    heads[1].view(3, 3, 25, 7, 7)[1, :, 4] = -9.0                        # ... nor on this one -> None
    _same_as_oracle(heads)


def test_yolo_nms_more_valid_candidates_than_the_lds_sort_holds():
    """19 200 candidates, all valid (> 16 384): the kernel's global-memory fallback path."""
    g = torch.Generator().manual_seed(69)
    heads = _clustered_heads(g, 1, 3, (80,), conf_shift=4.0, spread=15.0)
    _same_as_oracle(heads)


def test_greedy_prep_scores_and_labels_equal_the_max_over_sigmoids():
    """The prep kernel takes the row maximum on the logits and evaluates the sigmoid only on possible ties: scores and labels
    must equal `cls.sigmoid().max(1)` bit for bit, including saturated ties (sigmoid(20) == sigmoid(25) == 1 -> first index),
    near-ties within one float of the sigmoid, rows of equal logits and very negative rows."""
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    g = torch.Generator().manual_seed(12)
    An, C = 4096, 80
    cls = torch.randn(1, An, C, generator=g) * 3.0
    cls[0, 0, 5] = 20.0; cls[0, 0, 9] = 25.0                         # both saturate to 1.0: label 5
    cls[0, 1, 40] = 30.0; cls[0, 1, 3] = 17.5                        # 17.5 may or may not round to 1.0
    cls[0, 2, :] = 0.25                                              # all equal: label 0
    cls[0, 3, :] = -40.0; cls[0, 3, 7] = -39.9999
    base = torch.rand(An, generator=g) * 6
    cls[0, 4:, 11] = base[4:]; cls[0, 4:, 60] = base[4:] + (torch.rand(An - 4, generator=g) - 0.5) * 2e-4   # near ties around the max
    cls[0, 4:, 11] += 6.0; cls[0, 4:, 60] += 6.0
    cls[0, An - 1, :] = -120.0; cls[0, An - 1, 33] = -110.0          # 1 + exp(-l) overflows: every sigmoid is 0 -> label 0
    cls[0, An - 2, :] = -95.0; cls[0, An - 2, 20] = -85.0            # sigmoid(-85) = 1.2e-37 against zeros -> label 20
    cls[0, An - 3, 50] = float("nan"); cls[0, An - 3, 70] = float("nan")   # torch.max: NaN wins, the first one
    anchors = torch.cat([torch.rand(An, 2, generator=g) * 500 + 50, torch.rand(An, 2, generator=g) * 40 + 10], 1)
    loc = torch.zeros(1, An, 4)
    score_ref, label_ref = cls[0].cuda().sigmoid().max(1)            # the same device sigmoid family torch would use
    import ctypes
    from objectdetectionpl_amd import _lib
    L = _lib.lib()
    d = _lib.GreedyNmsDesc()
    locd, clsd, and_ = loc.cuda().contiguous(), cls.cuda().contiguous(), anchors.cuda().contiguous()
    d.loc, d.cls, d.anchors = locd.data_ptr(), clsd.data_ptr(), and_.data_ptr()
    d.B, d.An, d.C, d.topk = 1, An, C, 100
    d.nms_thresh, d.class_thresh = 0.5, 0.45
    need = L.odpl_greedy_nms_workspace_bytes(ctypes.byref(d))
    ws = torch.zeros(need, dtype=torch.uint8, device="cuda")
    out = torch.empty(1, 100, 7, device="cuda"); counts = torch.empty(1, 2, dtype=torch.int32, device="cuda")
    _lib.check(L.odpl_greedy_nms(ctypes.byref(d), _lib.ptr(ws), need, _lib.ptr(out), _lib.ptr(counts), _lib.current_stream(out.device)), "odpl_greedy_nms")
    torch.cuda.synchronize()
    score = ws[:An * 4].view(torch.float32)                           # workspace layout: score [B*An] then label [B*An] (256-B aligned)
    off = (An * 4 + 255) // 256 * 256
    label = ws[off:off + An * 4].view(torch.int32)
    cpu_ref_s, cpu_ref_l = cls[0].sigmoid().max(1)
    lab = label.cpu().long()
    assert torch.allclose(score.cpu(), cpu_ref_s, rtol=1e-6, atol=1e-7, equal_nan=True)
    assert int(lab[0]) == 5 and int(lab[2]) == 0 and int(lab[3]) == 7 and int(lab[1]) in (3, 40)
    assert int(lab[An - 1]) == 0 and int(lab[An - 2]) == 20 and int(lab[An - 3]) == 50
    assert float(score[An - 1]) == 0.0 and bool(torch.isnan(score[An - 3])) and torch.equal(lab[-3:], cpu_ref_l[-3:])
    top2 = cls[0].topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 0.05                          # unambiguous rows: the label is the arg-max of the logits
    clear[:4] = False; clear[-3:] = False
    assert int(clear.sum()) == 0 or torch.equal(lab[clear], cls[0].argmax(1)[clear])
    near = ~clear; near[:4] = False; near[-3:] = False
    picked = cls[0].gather(1, lab[:, None])[:, 0]                       # near ties: the label is one of the contenders within the
    assert bool((torch.sigmoid(picked[near]) >= torch.sigmoid(top2[near, 0]) - 1.2e-7).all())   # ... whose sigmoid ties with the maximum
    assert label_ref.shape == lab.shape


def test_yolo_merge_nms_at_config5_size_vs_oracle():
    """BASELINE config 5's NMS size: 3 x (80^2 + 40^2 + 20^2) = 25 200 candidates per image, 85 answers each, through the v5
    memory reinterpretation (YOLOV5.py:176-182), vs the oracle's sequential loop (kept set, order and merged boxes)."""
    from objectdetectionpl_amd.LightningFunc.nms import yolo_non_max_suppression
    from oracle import nms as oracle_nms
    g = torch.Generator().manual_seed(77)
    B, C = 2, 80
    heads = []
    for G in (80, 40, 20):
        h = torch.randn(B, 3, 85, G, G, generator=g)                 # what the NMS reads the [B,3,G,G,85] memory as
        h[:, :, 0:2] = torch.rand(B, 3, 2, 1, 1, generator=g) * 400 + 100 + torch.randn(B, 3, 2, G, G, generator=g) * 60
        h[:, :, 2:4] = torch.rand(B, 3, 2, G, G, generator=g) * 90 + 25
        h[:, :, 4] = torch.randn(B, 3, G, G, generator=g) * 0.6 - 1.6   # ~ 1 % above the -0.0151 threshold
        h[:, :, 5:] = torch.rand(B, 3, C, G, G, generator=g)
        heads.append(h.contiguous().view(B, 3, G, G, 85))            # same memory, v5's logical shape
    ref = oracle_nms.yolo_weighted_nms([h.reshape(B, 3 * 85, h.shape[2], h.shape[2]).clone() for h in heads])
    got = yolo_non_max_suppression([h.cuda() for h in heads])
    assert sum(h.shape[1] * h.shape[2] * h.shape[3] for h in heads) == 25200
    for r, o in zip(ref, got):
        assert (r is None) == (o is None)
        if r is not None:
            assert o.shape == r.shape and r.shape[0] > 50
            assert torch.equal(o[:, 6].cpu(), r[:, 6])                                  # labels: index work
            assert torch.allclose(o.cpu(), r, rtol=1e-4, atol=1e-3)


def test_greedy_nms_ties_at_the_topk_threshold_take_the_first_in_index_order():
    """[r4] the k-th largest score is found by a radix select on the float bits; when more candidates TIE at that score than fit into
    top-k, the reference's stable descending sort keeps the first of them in index order (RetinaNet.py:140-141).  6 000 anchors, 40 with
    distinct high scores, the rest all at the same logit (one score, thousands of ties), scattered boxes; and a second image where the
    ties exactly fit."""
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    g = torch.Generator().manual_seed(77)
    An, C = 6000, 6
    anchors = torch.cat([torch.rand(An, 2, generator=g) * 2000 + 50, torch.rand(An, 2, generator=g) * 30 + 10], 1)
    loc = torch.randn(2, An, 4, generator=g) * 0.05
    cls = torch.full((2, An, C), -6.0)
    cls[0, :, 2] = 0.75                                               # every anchor: the same score above the threshold
    hi = torch.randperm(An, generator=g)[:40]
    cls[0, hi, 4] = 1.0 + torch.rand(40, generator=g)                 # 40 distinct better ones
    cls[1, :, :] = -6.0
    pick = torch.randperm(An, generator=g)[:100]
    cls[1, pick[:70], 1] = 1.0 + torch.rand(70, generator=g)          # 70 distinct ...
    cls[1, pick[70:], 3] = 0.5                                         # ... + 30 ties that exactly fill top-100
    ref = oracle_retina.greedy_nms(loc, cls, anchors, topk=100, nms_thresh=0.5, class_thresh=0.45)
    out = greedy_non_max_suppression((loc.cuda(), cls.cuda()), anchors.cuda())
    for r, o in zip(ref, out):
        o = o.cpu()
        assert o.shape == r.shape and o.shape[0] > 20
        assert torch.equal(o[:, 4], r[:, 4]) and torch.equal(o[:, 6], r[:, 6])                   # the same candidates, in the same order
        assert torch.allclose(o[:, 5], r[:, 5], rtol=1e-6, atol=1e-7)                             # (device vs host expf in the sigmoid)
        assert torch.allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)


def test_greedy_nms_longer_topk_takes_the_sweep_loop():
    """topk = 300 (> 128 survivors): the workgroup-wide sort and the sweep loop instead of one wave's overlap matrix; vs the oracle"""
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    g = torch.Generator().manual_seed(78)
    An, C = 9000, 5
    anchors = torch.cat([torch.rand(An, 2, generator=g) * 600 + 50, torch.rand(An, 2, generator=g) * 60 + 20], 1)
    loc = torch.randn(2, An, 4, generator=g) * 0.1
    cls = torch.randn(2, An, C, generator=g) * 1.5 - 1.0
    ref = oracle_retina.greedy_nms(loc, cls, anchors, topk=300, nms_thresh=0.5, class_thresh=0.45)
    out = greedy_non_max_suppression((loc.cuda(), cls.cuda()), anchors.cuda(), topk=300)
    for r, o in zip(ref, out):
        o = o.cpu()
        assert o.shape == r.shape and o.shape[0] > 50
        assert torch.equal(o[:, 4], r[:, 4]) and torch.equal(o[:, 6], r[:, 6])
        assert torch.allclose(o[:, 5], r[:, 5], rtol=1e-6, atol=1e-7) and torch.allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)


def test_greedy_nms_short_topk_with_ties_inside_the_key_list():
    """[r4] the short way to the top 100 (bound from the threads' maxima -> LDS list of the keys at or above it -> order by counting):
    30 000 candidates above the threshold on one image -- 60 distinct high scores, 200 candidates TIED at one score below them, everything
    else lower and distinct.  The top 100 are the 60 + the first 40 of the ties in index order (RetinaNet.py:140-141: a stable descending
    sort), in that order; vs the oracle.  A second image: 5 000 candidates, all distinct."""
    from objectdetectionpl_amd.LightningFunc.nms import greedy_non_max_suppression
    g = torch.Generator().manual_seed(123)
    An, C = 30000, 4
    anchors = torch.cat([torch.rand(An, 2, generator=g) * 3000 + 50, torch.rand(An, 2, generator=g) * 30 + 10], 1)
    loc = torch.randn(2, An, 4, generator=g) * 0.05
    cls = torch.full((2, An, C), -6.0)
    cls[0, :, 1] = torch.rand(An, generator=g) * 1.5                   # everyone above 0.45 (sigmoid(0) = 0.5), distinct, below 0.82
    perm = torch.randperm(An, generator=g)
    cls[0, perm[:200], 2] = 2.0                                        # 200 ties at sigmoid(2.0) = 0.88
    cls[0, perm[200:260], 3] = 3.0 + torch.rand(60, generator=g)       # 60 distinct better ones
    cls[1, :5000, 0] = torch.rand(5000, generator=g) * 4.0
    ref = oracle_retina.greedy_nms(loc, cls, anchors, topk=100, nms_thresh=0.5, class_thresh=0.45)
    out = greedy_non_max_suppression((loc.cuda(), cls.cuda()), anchors.cuda())
    for r, o in zip(ref, out):
        o = o.cpu()
        assert o.shape == r.shape and o.shape[0] > 50
        assert torch.equal(o[:, 4], r[:, 4]) and torch.equal(o[:, 6], r[:, 6])
        assert torch.allclose(o[:, 5], r[:, 5], rtol=1e-6, atol=1e-7)
        assert torch.allclose(o[:, :4], r[:, :4], rtol=1e-5, atol=1e-3)
