"""The loss-driven training step of RetinaNet (csrc/odpl_sparsefwd.hip, model/RetinaNet.py:forward_and_loss): assignment first,
then the heads evaluated only where RetinaNetLoss (losses.py:405-472) will read -- against the dense forward + criterion."""
import argparse
import ctypes
import math

import numpy as np
import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

from objectdetectionpl_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol=1e-4, atol_scale=1e-5):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    assert a.shape == b.shape
    tol = atol_scale * float(b.abs().max()) + 1e-12
    assert torch.allclose(a, b, rtol=rtol, atol=tol), float((a - b).abs().max())


def _targets(g, B, per, C):
    n = B * per
    wh = torch.rand(n, 2, generator=g) * 0.35 + 0.15
    cxy = torch.rand(n, 2, generator=g) * (1 - wh) + wh / 2
    return torch.cat([torch.arange(B).repeat_interleave(per).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(), cxy, wh], 1)


@pytest.mark.parametrize("B,shapes,depth,mask,npos", [(2, [(16, 20), (8, 10), (4, 5)], 5, 0b011, 40), (3, [(33, 7), (5, 5)], 3, 0b11, 9),
                                                      (1, [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)], 5, 0b00101, 300),
                                                      (2, [(9, 9)], 8, 1, 0)])
def test_pixel_support_lists_are_the_dilated_positive_pixels(B, shapes, depth, mask, npos):
    """slot (l, j) = the pixels of level l within Chebyshev distance j of a pixel owning a positive anchor (numpy dilation); list
    order is not defined, so sets are compared; levels off the mask get nothing."""
    from objectdetectionpl_amd.LightningFunc.utils.retinahead import PixelSupport
    A = 9
    An = A * sum(h * w for h, w in shapes)
    rng = np.random.default_rng(3)
    rows = np.sort(rng.choice(B * An, size=npos, replace=False)).astype(np.int32)
    pos = torch.zeros(B * An, dtype=torch.int32, device=DEV)
    pos[:npos] = torch.from_numpy(rows).to(DEV)
    cnt = torch.tensor([npos], dtype=torch.int32, device=DEV)
    sup = PixelSupport(pos, cnt, B, shapes, depth, mask)
    torch.cuda.synchronize()
    first = np.cumsum([0] + [A * h * w for h, w in shapes])
    for l, (h, w) in enumerate(shapes):
        m0 = np.zeros((B, h, w), dtype=bool)
        for r in rows:
            b, a = divmod(int(r), An)
            if first[l] <= a < first[l + 1]:
                pix = (a - first[l]) // A
                m0[b, pix // w, pix % w] = True
        for j in range(depth):
            lst, c = sup.pair(l, j)
            n = int(c.item())
            if not (mask >> l) & 1:
                assert n == 0
                continue
            mj = np.zeros_like(m0)
            for dy in range(-j, j + 1):
                for dx in range(-j, j + 1):
                    sh = np.zeros_like(m0)
                    ys, xs = slice(max(dy, 0), h + min(dy, 0)), slice(max(dx, 0), w + min(dx, 0))
                    yd, xd = slice(max(-dy, 0), h + min(-dy, 0)), slice(max(-dx, 0), w + min(-dx, 0))
                    sh[:, yd, xd] = m0[:, ys, xs]
                    mj |= sh
            want = np.flatnonzero(mj.reshape(-1))
            got = np.sort(lst[:n].cpu().numpy())
            assert n == len(want) and np.array_equal(got, want), (l, j)


@pytest.mark.parametrize("B,H,W,cin,cout,act,weights_cl,frac", [(2, 20, 24, 64, 64, "relu", True, 0.2), (1, 7, 9, 32, 36, "linear", False, 0.5),
                                                                (2, 16, 16, 256, 720, "linear", True, 0.1), (1, 12, 12, 96, 100, "leaky", True, 1.0),
                                                                (2, 10, 10, 64, 128, "relu", True, 0.0)])
def test_forward_on_pixels_is_the_dense_convolution_there_and_nothing_elsewhere(B, H, W, cin, cout, act, weights_cl, frac):
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(5)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(cin, cout, 3, 1, 1).to(DEV)
    if weights_cl:
        conv = conv.to(memory_format=torch.channels_last)
    P = B * H * W
    sel = torch.randperm(P, generator=g)[:int(round(frac * P))].to(torch.int32)
    lst = torch.full((P,), -7, dtype=torch.int32, device=DEV)
    lst[:sel.numel()] = sel.to(DEV)
    cnt = torch.tensor([sel.numel()], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        want = F.conv2d(x, conv.weight, conv.bias, padding=1)
        want = {"relu": F.relu, "linear": lambda t: t, "leaky": lambda t: F.leaky_relu(t, 0.1)}[act](want)
        got = sc.forward_on_pixels(x, conv.weight, conv.bias, sc.ACT[act], (lst, cnt))
    on = torch.zeros(P, dtype=torch.bool)
    on[sel.long()] = True
    gm = got.permute(0, 2, 3, 1).reshape(P, cout).cpu()
    wm = want.permute(0, 2, 3, 1).reshape(P, cout).cpu()
    assert torch.count_nonzero(gm[~on]) == 0                                   # the zeros the caller put there
    if on.any():
        _close(gm[on], wm[on], 1e-4, 2e-6)


def test_forward_on_pixels_writes_prediction_table_rows():
    """the final convolutions write straight into a level's slice of the [B, An, no] table (batch stride An * no, pixel stride 9 * no)"""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(6)
    B, H, W, cin, no = 2, 6, 5, 32, 4
    An = 9 * (H * W + 11)
    x = torch.randn(B, cin, H, W, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    conv = nn.Conv2d(cin, 9 * no, 3, 1, 1).to(DEV)
    table = torch.zeros(B, An, no, device=DEV)
    first = 9 * 11
    P = B * H * W
    sel = torch.arange(0, P, 3, dtype=torch.int32, device=DEV)
    lst = torch.zeros(P, dtype=torch.int32, device=DEV)                        # (a list has room for every pixel of its map)
    lst[:sel.numel()] = sel
    cnt = torch.tensor([sel.numel()], dtype=torch.int32, device=DEV)
    with torch.no_grad():
        sc.forward_on_pixels(x, conv.weight, conv.bias, 0, (lst, cnt), out=table[:, first:first + 9 * H * W], out_strides=(An * no, 9 * no))
        want = conv(x).permute(0, 2, 3, 1).reshape(B, H * W * 9, no)
    got = table[:, first:first + 9 * H * W].reshape(B * H * W, 9 * no).cpu()
    want = want.reshape(B * H * W, 9 * no).cpu()
    on = torch.zeros(P, dtype=torch.bool)
    on[sel.long().cpu()] = True
    _close(got[on], want[on], 1e-4, 2e-6)
    assert torch.count_nonzero(got[~on]) == 0 and torch.count_nonzero(table[:, :first]) == 0
    with pytest.raises(ValueError):
        sc.forward_on_pixels(x, conv.weight, conv.bias, 0, (sel, cnt))


def test_split_assignment_gives_the_one_call_loss_bit_for_bit():
    """ODPL_RETINA_ASSIGN_ONLY, then ODPL_RETINA_ASSIGNED on the same workspace == one odpl_retina_loss call: loss, positive rows, the
    gradient on those rows"""
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    g = torch.Generator().manual_seed(7)
    B, C, size = 3, 6, 192
    anchors = get_anchor_boxes(torch.Tensor([size, size]), torch.device(DEV))
    An = anchors.shape[0]
    loc = torch.randn(B, An, 4, generator=g).to(DEV) * 0.3
    cls = torch.randn(B, An, C, generator=g).to(DEV)
    y = _targets(g, B, 4, C).to(DEV)
    crit = RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, size)
    res = []
    for split in (False, True):
        g_loc, g_cls = torch.zeros_like(loc), torch.zeros_like(cls)
        pos = torch.zeros(B * An, dtype=torch.int32, device=DEV)
        npos = torch.zeros(1, dtype=torch.int32, device=DEV)
        if split:
            pos, npos, _ = crit.assign(y, B)
            first = (pos.clone(), npos.clone())
            out3, counts = crit._launch(loc, cls, g_loc, g_cls, y, rows_only=(pos, npos), mode="assigned")
            assert torch.equal(first[0][:int(first[1])], pos[:int(npos)]) and torch.equal(first[1], npos)
        else:
            out3, counts = crit._launch(loc, cls, g_loc, g_cls, y, rows_only=(pos, npos))
        res.append((out3.clone(), counts.clone(), pos[:int(npos)].clone(), g_loc, g_cls))
    assert int(res[0][1][0]) > 10
    for a, b in zip(res[0], res[1]):
        assert torch.equal(a, b)
    with pytest.raises(_lib.OdplError):                                          # nothing promised any more
        crit._launch(loc, cls, torch.zeros_like(loc), torch.zeros_like(cls), y, rows_only=(pos, npos), mode="assigned")


def _model(C, size):
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    args = argparse.Namespace(model_name="RetinaNet", cls_criterion="focal_loss", coord_criterion="smooth_l1_loss", data_module="synthetic")
    RetinaNet.img_size = size
    try:
        return RetinaNet([str(i) for i in range(C)], args).to(DEV).to(memory_format=torch.channels_last).train()
    finally:
        RetinaNet.img_size = 600


class _EveryPixel:
    """a PixelSupport whose every list is the whole map: the same kernels, evaluated everywhere"""

    def __init__(self, B, shapes, mask):
        self.mask = mask
        self.lists = [torch.arange(B * h * w, dtype=torch.int32, device=DEV) for h, w in shapes]
        self.counts = [torch.tensor([B * h * w], dtype=torch.int32, device=DEV) for h, w in shapes]

    def has(self, l):
        return bool((self.mask >> l) & 1)

    def pair(self, l, j):
        return self.lists[l], self.counts[l]


def _heads_run(model, feats, y, B, mode, mask):
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    fs = [f.clone().requires_grad_() for f in feats]
    model.zero_grad(set_to_none=True)
    crit = model.criterion
    if mode == "dense":
        out = rh.final_convs(model.loc_head[-1], model.cls_head[-1], [model._run_tower(model.loc_head, fm) for fm in fs],
                             [model._run_tower(model.cls_head, fm) for fm in fs])
    else:
        pos, npos, token = crit.assign(y, B)
        shapes = [(f.shape[2], f.shape[3]) for f in fs]
        sup = rh.PixelSupport(pos, npos, B, shapes, 5, mask) if mode == "driven" else _EveryPixel(B, shapes, mask)
        xl = [model._run_tower(model.loc_head, fm, sup, l) for l, fm in enumerate(fs)]
        xc = [model._run_tower(model.cls_head, fm, sup, l) for l, fm in enumerate(fs)]
        out = rh.final_convs(model.loc_head[-1], model.cls_head[-1], xl, xc, support=sup, assignment=token)
    d = crit(out, y)
    d["loss"].backward()
    n = int(crit.last_counts[0])
    return dict(loss={k: v.detach().clone() for k, v in d.items()}, out=[o.detach().clone() for o in out], gf=[f.grad.clone() for f in fs],
                gp={k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, n=n)


@pytest.mark.parametrize("size,B,mask", [(256, 2, 0b11111), (320, 3, 0b11111), (192, 1, 0b11111)])      # (a level off the mask runs the
def test_restricting_the_heads_to_the_loss_pixels_changes_no_bit(size, B, mask):                           # library, not reproducible run to run)
    """Both heads on fixed feature maps, the same kernels twice: once on the pixel lists of `odpl_pixel_support`, once on EVERY pixel.
    Unit t is only read where unit t + 1 / the criterion looks and the backward walks subsets of the same sets, so the loss dict,
    the predictions on the positive rows, d loss / d (every head parameter) and d loss / d (every feature map) are equal bit for bit."""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    torch.manual_seed(0)
    C = 5
    model = _model(C, size)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, B, 3, C).to(DEV)
    with torch.no_grad():
        feats = [f.clone() for f in model.fpn(x)]
    old = sc.min_pixels, sc.max_density
    sc.min_pixels, sc.max_density = 1, 2.0
    try:
        full = _heads_run(model, feats, y, B, "everywhere", mask)
        pos, npos, _ = model.criterion.assign(y, B)
        rows = pos[:int(npos)].long().clone()
        drv = _heads_run(model, feats, y, B, "driven", mask)
    finally:
        sc.min_pixels, sc.max_density = old
    assert full["n"] == drv["n"] == rows.numel() > 20
    for k in full["loss"]:
        assert torch.equal(full["loss"][k], drv["loss"][k]), k
    assert torch.equal(drv["out"][0].reshape(-1, 4)[rows], full["out"][0].reshape(-1, 4)[rows])
    assert torch.equal(drv["out"][1].reshape(-1, C)[rows], full["out"][1].reshape(-1, C)[rows])
    assert len(full["gp"]) == len(drv["gp"]) == 20
    for k in full["gp"]:
        assert torch.equal(full["gp"][k], drv["gp"][k]), k
    for a, b in zip(drv["gf"], full["gf"]):
        assert torch.equal(a, b)


def _rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / (b.norm() + 1e-300))


@pytest.mark.parametrize("size,B,mask", [(256, 2, 0b11111), (320, 3, 0b00001), (256, 1, 0b00110)])
def test_heads_on_the_loss_pixels_against_the_library_heads(size, B, mask):
    """The same against the dense heads (MIOpen convolutions + the fused epilogues): loss 1e-5, predictions on the positive rows 1e-4;
    gradients in relative L2 -- a ReLU whose pre-activation the two forward kernels round to different sides of zero moves single
    gradient elements by more than rounding (the library shows the same scatter against itself from run to run)."""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    torch.manual_seed(0)
    C = 5
    model = _model(C, size)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, B, 3, C).to(DEV)
    with torch.no_grad():
        feats = [f.clone() for f in model.fpn(x)]
    old = sc.min_pixels, sc.max_density
    sc.min_pixels, sc.max_density = 1, 2.0
    try:
        dn = _heads_run(model, feats, y, B, "dense", mask)
        pos, npos, _ = model.criterion.assign(y, B)
        rows = pos[:int(npos)].long().clone()
        dr = _heads_run(model, feats, y, B, "driven", mask)
    finally:
        sc.min_pixels, sc.max_density = old
    assert dn["n"] == dr["n"] > 20
    for k in dn["loss"]:
        assert abs(float(dn["loss"][k]) - float(dr["loss"][k])) <= 1e-5 * abs(float(dn["loss"][k])), k
    _close(dr["out"][0].reshape(-1, 4)[rows], dn["out"][0].reshape(-1, 4)[rows], 1e-4, 1e-5)
    _close(dr["out"][1].reshape(-1, C)[rows], dn["out"][1].reshape(-1, C)[rows], 1e-4, 1e-5)
    for k in dn["gp"]:
        assert _rel_l2(dr["gp"][k], dn["gp"][k]) < 5e-2, (k, _rel_l2(dr["gp"][k], dn["gp"][k]))
    for a, b in zip(dr["gf"], dn["gf"]):
        assert _rel_l2(a, b) < 5e-2 or float(b.abs().max()) == 0.0


def test_training_step_takes_the_loss_driven_route():
    """`training_step` (step.py:14-25) end to end on both routes: same loss dict and positive count, every parameter gets a gradient,
    and the two full gradients point the same way (cosine; element-wise equality is not defined across two runs of the backbone)."""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    torch.manual_seed(0)
    C, size, B = 5, 384, 3
    model = _model(C, size)
    g = torch.Generator().manual_seed(11)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, B, 3, C).to(DEV)
    old = sc.min_pixels
    sc.min_pixels = 1500                                                         # levels 48^2 and 24^2 x 3 images on pixel lists
    res = []
    try:
        for driven in (True, False):
            RetinaNet.loss_driven_step = driven
            model.zero_grad(set_to_none=True)
            calls = []
            orig = sc.forward_on_pixels
            sc.forward_on_pixels = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
            try:
                d = model.training_step((x, y), 0)
            finally:
                sc.forward_on_pixels = orig
            d["loss"].backward()
            assert len(calls) == (24 if driven else 0)      # 2 levels x (2 heads x (4 tower units + the final convolution) + FPN lateral + smoothing)
            assert all(p.grad is not None for p in model.parameters())
            res.append(({k: float(v.detach()) for k, v in d.items()}, torch.cat([p.grad.flatten().double() for p in model.parameters()]),
                        int(model.criterion.last_counts[0])))
    finally:
        RetinaNet.loss_driven_step = True
        sc.min_pixels = old
    assert res[0][2] == res[1][2] > 0
    for k in res[1][0]:
        assert abs(res[0][0][k] - res[1][0][k]) <= 2e-5 * abs(res[1][0][k]), (k, res[0][0], res[1][0])
    cos = float(torch.dot(res[0][1], res[1][1]) / (res[0][1].norm() * res[1][1].norm()))
    assert cos > 0.999, cos            # (two dense runs of the same step: 0.99990-0.99995; a missing level or tap: < 0.99)


def test_predictions_of_a_loss_driven_step_serve_that_assignment_only():
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    torch.manual_seed(0)
    model = _model(3, 128)
    g = torch.Generator().manual_seed(12)
    x = torch.rand(2, 3, 128, 128, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, 2, 2, 3).to(DEV)
    crit = model.criterion
    old = sc.min_pixels
    sc.min_pixels = 1
    try:
        feats = model.fpn(x)
        pos, npos, token = crit.assign(y, 2)
        sup = rh.PixelSupport(pos, npos, 2, [(f.shape[2], f.shape[3]) for f in feats], 5, 0b00011)
        xl = [model._run_tower(model.loc_head, f, sup, l) for l, f in enumerate(feats)]
        xc = [model._run_tower(model.cls_head, f, sup, l) for l, f in enumerate(feats)]
        out = rh.final_convs(model.loc_head[-1], model.cls_head[-1], xl, xc, support=sup, assignment=token)
        crit.assign(y, 2)                                                     # a newer assignment replaces the one these rows were computed for
        with pytest.raises(_lib.OdplError, match="one RetinaNetLoss.assign call only"):
            crit(out, y)
    finally:
        sc.min_pixels = old


def test_eval_no_grad_and_small_inputs_decline_before_running_anything():
    model = _model(3, 128)
    g = torch.Generator().manual_seed(13)
    x = torch.rand(2, 3, 128, 128, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, 2, 2, 3).to(DEV)
    assert model.loss_driven_mask(x, y) == 0 and model.forward_and_loss(x, y) is None       # 128^2: every level is below sparseconv.min_pixels
    d = model.training_step((x, y), 0)                                                     # ... and the step is forward + criterion
    assert math.isfinite(float(d["loss"].detach()))
    big = torch.rand(2, 3, 512, 512, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    assert model.loss_driven_mask(big, y) == 0                                             # the criterion's anchor table is for 128^2
    model2 = _model(3, 512)
    assert model2.loss_driven_mask(big, y) == 0b00001
    with torch.no_grad():
        assert model2.loss_driven_mask(big, y) == 0
    with torch.autocast("cuda", dtype=torch.bfloat16):
        assert model2.loss_driven_mask(big, y) == 0
    model2.eval()
    assert model2.loss_driven_mask(big, y) == 0


@pytest.mark.parametrize("hw", [(800, 1333), (257, 300), (64, 96), (130, 129)])
def test_level_shapes_are_what_the_fpn_produces(hw):
    model = _model(3, 128)
    x = torch.rand(1, 3, *hw).to(DEV).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        feats = model.fpn(x)
    assert [(f.shape[2], f.shape[3]) for f in feats] == model._level_shapes(x)


def test_config4_size_restriction_changes_no_bit():
    """BASELINE config 4 (B = 4, 800 x 1333 -> 200 700 anchors, C = 80, 256-channel features) on random feature maps: the heads on the
    criterion's pixels (the two large levels on lists, as the step runs them) against the same kernels on every pixel of those levels."""
    import math
    from objectdetectionpl_amd.LightningFunc.losses_dense import RetinaNetLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils.RetinaUtils import get_anchor_boxes
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    torch.manual_seed(0)
    C, B, H, W = 80, 4, 800, 1333
    model = _model(C, 128)
    anchors = get_anchor_boxes(torch.Tensor([W, H]), torch.device(DEV))
    assert anchors.shape[0] == 200700
    model.iou_boxes, model.criterion = anchors, RetinaNetLoss(anchors, focal_loss, "smooth_l1_loss", C, (W, H))
    g = torch.Generator().manual_seed(21)
    shapes = [(math.ceil(H / 2 ** (3 + l)), math.ceil(W / 2 ** (3 + l))) for l in range(5)]
    feats = [(torch.randn(B, h, w, 256, generator=g) * 0.5).to(DEV).permute(0, 3, 1, 2) for h, w in shapes]
    n = B * 8
    wh = torch.rand(n, 2, generator=g) * 0.45 + 0.05
    y = torch.cat([torch.arange(B).repeat_interleave(8).float()[:, None], torch.randint(0, C, (n, 1), generator=g).float(),
                   torch.rand(n, 2, generator=g) * 0.6 + 0.2, wh], 1).to(DEV)
    mask = 0b00011
    assert model.loss_driven_mask(torch.empty(B, 3, H, W, device=DEV), y) == mask
    old = sc.max_density
    sc.max_density = 2.0
    try:
        full = _heads_run(model, feats, y, B, "everywhere", mask)
        pos, npos, _ = model.criterion.assign(y, B)
        rows = pos[:int(npos)].long().clone()
        drv = _heads_run(model, feats, y, B, "driven", mask)
    finally:
        sc.max_density = old
    assert full["n"] == drv["n"] == rows.numel() > 300
    first_small = 9 * (shapes[0][0] * shapes[0][1] + shapes[1][0] * shapes[1][1])        # rows of P5-P7 come from the library in both runs
    big = rows[(rows % anchors.shape[0]) < first_small]
    assert big.numel() > 100
    assert torch.equal(drv["out"][0].reshape(-1, 4)[big], full["out"][0].reshape(-1, 4)[big]), "loc rows"
    assert torch.equal(drv["out"][1].reshape(-1, C)[big], full["out"][1].reshape(-1, C)[big]), "cls rows"
    for i, (a, b) in enumerate(zip(drv["gf"][:2], full["gf"][:2])):                        # d loss / d P3, P4: the listed levels only
        assert torch.equal(a, b), (i, float((a - b).abs().max()))
    for k in full["loss"]:
        assert abs(float(full["loss"][k]) - float(drv["loss"][k])) <= 1e-6 * abs(float(full["loss"][k]))
    for k in full["gp"]:                                                                 # (sums over all five levels: P5-P7 run the library in both
        assert _rel_l2(drv["gp"][k], full["gp"][k]) < 5e-2, (k, _rel_l2(drv["gp"][k], full["gp"][k]))   # runs, not reproducible, ReLU flips)


# ---- the FPN's top-down path under the same restriction ----------------------------------------------------------------------

@pytest.mark.parametrize("B,H,W,Hf,Wf,na,nb", [(2, 13, 17, 25, 34, 9, 30), (1, 50, 84, 100, 167, 0, 400), (3, 8, 8, 16, 16, 20, 0), (2, 5, 6, 10, 11, 0, 0)])
def test_union_lists_mark_own_pixels_and_bilinear_sources(B, H, W, Hf, Wf, na, nb):
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    rng = np.random.default_rng(4)
    own = np.sort(rng.choice(B * H * W, size=na, replace=False)).astype(np.int32)
    fin = np.sort(rng.choice(B * Hf * Wf, size=nb, replace=False)).astype(np.int32)

    def dev_list(a, n):
        t = torch.zeros(n, dtype=torch.int32, device=DEV)
        t[:len(a)] = torch.from_numpy(a).to(DEV)
        return t, torch.tensor([len(a)], dtype=torch.int32, device=DEV)
    if na == 0 and nb == 0:
        with pytest.raises(_lib.OdplError):
            sc.union_lists(B, H, W)                                            # nothing to build from
        return
    (marked, cm), (near, cn) = sc.union_lists(B, H, W, own=dev_list(own, B * H * W) if na else None,
                                             finer=dev_list(fin, B * Hf * Wf) if nb else None, finer_hw=(Hf, Wf))
    m = np.zeros((B, H, W), dtype=bool)
    m.reshape(-1)[own] = True
    for q in fin:
        b, rem = divmod(int(q), Hf * Wf)
        y, x = divmod(rem, Wf)
        fy = max(np.float32(H / Hf) * np.float32(y + 0.5) - np.float32(0.5), np.float32(0))
        fx = max(np.float32(W / Wf) * np.float32(x + 0.5) - np.float32(0.5), np.float32(0))
        y0, x0 = int(fy), int(fx)
        y1, x1 = y0 + (1 if y0 < H - 1 else 0), x0 + (1 if x0 < W - 1 else 0)
        for yy in (y0, y1):
            for xx in (x0, x1):
                m[b, yy, xx] = True
    n = np.zeros_like(m)
    for dy in (-1, 0, 1):
        for dx in (-1, 0, 1):
            sh = np.zeros_like(m)
            sh[:, max(-dy, 0):H + min(-dy, 0), max(-dx, 0):W + min(-dx, 0)] = m[:, max(dy, 0):H + min(dy, 0), max(dx, 0):W + min(dx, 0)]
            n |= sh
    assert np.array_equal(marked[:int(cm)].cpu().numpy(), np.flatnonzero(m.reshape(-1)))          # ascending
    assert np.array_equal(near[:int(cn)].cpu().numpy(), np.flatnonzero(n.reshape(-1)))


@pytest.mark.parametrize("B,C,h,w,H,W", [(2, 64, 13, 17, 25, 34), (1, 256, 50, 84, 100, 167), (2, 32, 4, 4, 8, 8), (1, 16, 3, 5, 3, 5)])
def test_up_add_on_pixels_is_interpolate_plus_lateral_there(B, C, h, w, H, W):
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(8)
    coarse = torch.randn(B, h, w, C, generator=g).to(DEV).permute(0, 3, 1, 2).requires_grad_()
    lat = torch.randn(B, H, W, C, generator=g).to(DEV).permute(0, 3, 1, 2).requires_grad_()
    P = B * H * W
    sel = torch.randperm(P, generator=g)[:max(1, P // 3)].to(torch.int32)
    lst = torch.zeros(P, dtype=torch.int32, device=DEV)
    lst[:sel.numel()] = sel.to(DEV)
    cnt = torch.tensor([sel.numel()], dtype=torch.int32, device=DEV)
    on = torch.zeros(P, dtype=torch.bool, device=DEV)
    on[sel.long().to(DEV)] = True
    onm = on.view(B, 1, H, W)
    got = sc.up_add_on_pixels(coarse, lat, (lst, cnt))
    want = F.interpolate(coarse, size=(H, W), mode="bilinear", align_corners=False) + lat
    _close(got, want * onm, 1e-5, 1e-6)
    assert torch.count_nonzero(got * (~onm)) == 0
    gy = torch.randn(B, H, W, C, generator=g).to(DEV).permute(0, 3, 1, 2) * onm
    gc, gl = torch.autograd.grad(got, [coarse, lat], gy)
    wc, wl = torch.autograd.grad(want, [coarse, lat], gy)
    _close(gc, wc, 1e-4, 1e-6)
    assert torch.equal(gl, wl)


@pytest.mark.parametrize("B,H,W,cin,cout,weights_cl,dens", [(2, 20, 24, 512, 256, True, 0.05), (1, 9, 7, 128, 64, False, 0.3), (2, 16, 16, 64, 64, True, 1.0)])
def test_one_by_one_convolution_on_pixels_forward_and_backward(B, H, W, cin, cout, weights_cl, dens):
    """the FPN laterals (1 x 1, pad 0) through the same kernels: forward on a list == F.conv2d there; backward from a sparse gradient ==
    aten::convolution_backward"""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    g = torch.Generator().manual_seed(9)
    x = torch.randn(B, H, W, cin, generator=g).to(DEV).permute(0, 3, 1, 2).requires_grad_()
    conv = nn.Conv2d(cin, cout, 1).to(DEV)
    if weights_cl:
        conv = conv.to(memory_format=torch.channels_last)
    P = B * H * W
    sel = torch.nonzero(torch.rand(P, generator=g) < dens).flatten().to(torch.int32)
    lst = torch.zeros(P, dtype=torch.int32, device=DEV)
    lst[:sel.numel()] = sel.to(DEV)
    cnt = torch.tensor([sel.numel()], dtype=torch.int32, device=DEV)
    on = torch.zeros(P, dtype=torch.bool, device=DEV)
    on[sel.long().to(DEV)] = True
    onm = on.view(B, 1, H, W)
    old = sc.min_pixels
    sc.min_pixels = 1
    try:
        assert sc._applicable(x, conv)
        got = sc.conv_bias_act(x, conv, "linear", (lst, cnt))
        want = conv(x)
        _close(got, want.detach() * onm, 1e-4, 2e-6)
        gy = torch.randn(B, H, W, cout, generator=g).to(DEV).permute(0, 3, 1, 2) * onm
        ga = torch.autograd.grad(got, [x, conv.weight, conv.bias], gy)
        gb = torch.autograd.grad(want, [x, conv.weight, conv.bias], gy)
    finally:
        sc.min_pixels = old
    for a, b in zip(ga, gb):
        _close(a, b, 1e-4, 2e-6)


def _topdown_run(model, c3, c4, p5, rest, y, B, mode, mask):
    """FPN top-down (laterals, merges, smoothing) + both heads + criterion on fixed backbone features"""
    from objectdetectionpl_amd.LightningFunc.utils import retinahead as rh
    leaves = [t.clone().requires_grad_() for t in (c3, c4, p5)]
    model.zero_grad(set_to_none=True)
    crit = model.criterion
    fpn = model.fpn
    shapes = [(t.shape[2], t.shape[3]) for t in (c3, c4, p5)] + [(t.shape[2], t.shape[3]) for t in rest]
    if mode == "library":
        p4 = fpn._smooth(fpn.toplayer1, fpn._up_add(leaves[2], fpn.latlayer2(leaves[1])))
        p3 = fpn._smooth(fpn.toplayer2, fpn._up_add(p4, fpn.latlayer3(leaves[0])))
        feats = [p3, p4, leaves[2]] + list(rest)
        out = rh.final_convs(model.loc_head[-1], model.cls_head[-1], [model._run_tower(model.loc_head, fm) for fm in feats],
                             [model._run_tower(model.cls_head, fm) for fm in feats])
    else:
        pos, npos, token = crit.assign(y, B)
        sup = rh.PixelSupport(pos, npos, B, shapes, 7, mask) if mode == "driven" else _EveryPixel(B, shapes, mask)
        p3, p4 = fpn._topdown_on_pixels(leaves[0], leaves[1], leaves[2], sup, 5)
        feats = [p3, p4, leaves[2]] + list(rest)
        xl = [model._run_tower(model.loc_head, fm, sup, l) for l, fm in enumerate(feats)]
        xc = [model._run_tower(model.cls_head, fm, sup, l) for l, fm in enumerate(feats)]
        out = rh.final_convs(model.loc_head[-1], model.cls_head[-1], xl, xc, support=sup, assignment=token)
    d = crit(out, y)
    d["loss"].backward()
    return dict(loss={k: v.detach().clone() for k, v in d.items()}, gl=[t.grad.clone() for t in leaves],
                gp={k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}, n=int(crit.last_counts[0]))


@pytest.mark.parametrize("size,B,mask", [(256, 2, 0b00011), (320, 1, 0b00001), (384, 2, 0b00011)])
def test_topdown_on_the_loss_pixels_changes_no_bit_and_matches_the_library(size, B, mask):
    """Fixed backbone features (c3, c4, p5, p6, p7): laterals + merges + smoothing + heads + criterion with the pixel lists, with EVERY
    pixel through the same kernels (bit for bit: loss, d loss / d (c3, c4, p5), every FPN top-down and head parameter), and through
    the library (loss 1e-5, gradients 5e-2 in relative L2: a handful of flipped ReLUs among few positives)."""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    torch.manual_seed(0)
    C = 5
    model = _model(C, size)
    g = torch.Generator().manual_seed(14)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, B, 3, C).to(DEV)
    fpn = model.fpn
    with torch.no_grad():
        c1 = F.max_pool2d(torch.relu(fpn.bn1(fpn.conv1(x))), 3, 2, 1)
        c3 = fpn.layer2(fpn.layer1(c1)); c4 = fpn.layer3(c3); c5 = fpn.layer4(c4)
        p6 = fpn.conv6(c5); p7 = fpn.conv7(F.relu(p6)); p5 = fpn.latlayer1(c5)
        c3, c4, p5, p6, p7 = (t.contiguous(memory_format=torch.channels_last).clone() for t in (c3, c4, p5, p6, p7))
    old = sc.min_pixels, sc.max_density
    sc.min_pixels, sc.max_density = 1, 2.0
    try:
        lib = _topdown_run(model, c3, c4, p5, (p6, p7), y, B, "library", mask)
        full = _topdown_run(model, c3, c4, p5, (p6, p7), y, B, "everywhere", mask)
        drv = _topdown_run(model, c3, c4, p5, (p6, p7), y, B, "driven", mask)
    finally:
        sc.min_pixels, sc.max_density = old
    assert lib["n"] == full["n"] == drv["n"] > 10
    topdown = ("fpn.latlayer2", "fpn.latlayer3", "fpn.toplayer1", "fpn.toplayer2")
    for k in full["loss"]:
        assert abs(float(full["loss"][k]) - float(drv["loss"][k])) <= 1e-6 * abs(float(full["loss"][k])), k
        assert abs(float(lib["loss"][k]) - float(drv["loss"][k])) <= 1e-5 * abs(float(lib["loss"][k])), k
    listed = [l for l in (0, 1) if (mask >> l) & 1]                              # a level off the mask runs the library: not reproducible
    for i in (0, 1):                                                            # d loss / d c3, c4
        if i in listed:
            assert torch.equal(full["gl"][i], drv["gl"][i]), i
        assert _rel_l2(drv["gl"][i], lib["gl"][i]) < 5e-2 and _rel_l2(drv["gl"][i], full["gl"][i]) < 5e-2
    assert _rel_l2(drv["gl"][2], full["gl"][2]) < 5e-2 and _rel_l2(drv["gl"][2], lib["gl"][2]) < 5e-2   # p5 also feeds its own (library) heads
    exact = ("fpn.latlayer3", "fpn.toplayer2") + (("fpn.latlayer2", "fpn.toplayer1") if 1 in listed else ())
    seen = 0
    for k in full["gp"]:
        if k.startswith(exact):
            assert torch.equal(full["gp"][k], drv["gp"][k]), k
        if k.startswith(topdown):
            seen += 1
        assert _rel_l2(drv["gp"][k], lib["gp"][k]) < 5e-2, (k, _rel_l2(drv["gp"][k], lib["gp"][k]))
    assert seen == 8


@pytest.mark.parametrize("case", ["no_positive_anchor", "image_without_boxes", "label_out_of_range"])
def test_edge_cases_follow_the_dense_step(case):
    """What RetinaNetLoss does at the edges (losses.py:431 raises on an image without boxes -- here: NaN + status bit; no positive anchor:
    zero loss over max(1, 0); a label >= C: status bit) is the same on the loss-driven route, and every parameter still gets a gradient."""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    torch.manual_seed(0)
    C, size, B = 4, 320, 2
    model = _model(C, size)
    g = torch.Generator().manual_seed(15)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    if case == "no_positive_anchor":          # boxes far smaller than the smallest anchor: IoU < 0.4 everywhere
        y = torch.tensor([[0, 1, 0.5, 0.5, 0.004, 0.004], [1, 2, 0.3, 0.6, 0.003, 0.005]], device=DEV)
    elif case == "image_without_boxes":
        y = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.3], [0, 2, 0.3, 0.6, 0.2, 0.4]], device=DEV)
    else:
        y = torch.tensor([[0, 1, 0.5, 0.5, 0.3, 0.3], [1, 9, 0.3, 0.6, 0.2, 0.4]], device=DEV)
    old = sc.min_pixels
    sc.min_pixels = 1000
    res = []
    try:
        for driven in (True, False):
            RetinaNet.loss_driven_step = driven
            model.zero_grad(set_to_none=True)
            assert (model.loss_driven_mask(x, y) != 0) == driven
            d = model.training_step((x, y), 0)
            d["loss"].backward()
            counts = model.criterion.last_counts.cpu().tolist()
            res.append(({k: float(v.detach()) for k, v in d.items()}, counts, [p.grad is not None for p in model.parameters()]))
    finally:
        RetinaNet.loss_driven_step = True
        sc.min_pixels = old
    (la, ca, ga), (lb, cb, gb) = res
    assert ca == cb and ga == gb and all(ga)
    for k in lb:
        assert (math.isnan(la[k]) and math.isnan(lb[k])) or abs(la[k] - lb[k]) <= 2e-5 * abs(lb[k]) + 1e-12, (k, la, lb)
    if case == "no_positive_anchor":
        assert ca[0] == 0 and la["loss"] == 0.0
    elif case == "image_without_boxes":
        assert ca[1] & 1 and math.isnan(la["loss"])
    else:
        assert ca[1] & 2


def test_a_crowded_level_goes_back_to_the_library_route():
    """Boxes on every part of the image put most of a level's pixels within reach of a positive anchor: the counters are read back
    without a sync and the level leaves the mask on a later step (and comes back after `reprobe_every` steps)."""
    import time
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    torch.manual_seed(0)
    C, size, B = 3, 256, 2
    model = _model(C, size)
    g = torch.Generator().manual_seed(16)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    n = 40
    cxy = torch.rand(B * n, 2, generator=g) * 0.8 + 0.1
    y = torch.cat([torch.arange(B).repeat_interleave(n).float()[:, None], torch.randint(0, C, (B * n, 1), generator=g).float(), cxy,
                   torch.full((B * n, 2), 0.2)], 1).to(DEV)                       # 51-pixel boxes: P3 / P4 anchors, all over the image
    old = sc.min_pixels, RetinaNet.reprobe_every
    sc.min_pixels, RetinaNet.reprobe_every = 500, 3
    try:
        assert model.loss_driven_mask(x, y) == 0b00011
        d = model.training_step((x, y), 0)
        d["loss"].backward()
        torch.cuda.synchronize(); time.sleep(0.05)
        m = model.loss_driven_mask(x, y)
        assert m != 0b00011 and (m & ~0b00011) == 0                              # at least one of the two levels was found crowded
        masks = [model.loss_driven_mask(x, y) for _ in range(3)]
        assert masks[-1] == 0b00011                                              # tried again after reprobe_every looks
    finally:
        sc.min_pixels, RetinaNet.reprobe_every = old


@pytest.mark.parametrize("driven", [True, False])
def test_overfitting_one_batch_brings_the_loss_down_on_either_route(driven):
    """system-level sanity: 40 Adam steps on one batch through `training_step` -- the loss has to fall by more than half whichever way the
    step is computed (and on the loss-driven route the step has to have taken it)"""
    from objectdetectionpl_amd.LightningFunc.utils import sparseconv as sc
    from objectdetectionpl_amd.model.RetinaNet import RetinaNet
    torch.manual_seed(0)
    C, size, B = 4, 256, 2
    model = _model(C, size)
    g = torch.Generator().manual_seed(17)
    x = torch.rand(B, 3, size, size, generator=g).to(DEV).contiguous(memory_format=torch.channels_last)
    y = _targets(g, B, 3, C).to(DEV)
    opt = torch.optim.Adam(model.parameters(), lr=1e-4)
    old = sc.min_pixels
    sc.min_pixels = 500
    RetinaNet.loss_driven_step = driven
    try:
        assert (model.loss_driven_mask(x, y) != 0) == driven
        losses = []
        for _ in range(40):
            opt.zero_grad(set_to_none=True)
            d = model.training_step((x, y), 0)
            d["loss"].backward()
            opt.step()
            losses.append(float(d["loss"].detach()))
    finally:
        RetinaNet.loss_driven_step = True
        sc.min_pixels = old
    assert all(math.isfinite(v) for v in losses)
    assert min(losses[-5:]) < 0.5 * losses[0], (losses[0], losses[-5:])
