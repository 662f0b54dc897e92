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


# ---- the structured backward of SSD's head pairs (utils/ssdhead.py; model/SSD.py:249-310, losses.py:159-315) ---------------------------

SSD_SOURCES = ((512, 38, 4), (1024, 19, 6), (512, 10, 6), (256, 5, 6), (256, 3, 4), (256, 1, 4))        # (Cin, H = W, priors per cell): 8 732 priors


def _ssd_targets(g, B, per_image, C):
    n = B * per_image
    wh = torch.rand(n, 2, generator=g) * 0.4 + 0.08
    cxy = torch.rand(n, 2, generator=g) * 0.6 + 0.2
    return torch.cat([torch.arange(B).repeat_interleave(per_image).float()[:, None], torch.randint(1, C, (n, 1), generator=g).float(), cxy, wh], 1)


def test_ssd_loss_rows_only_lists_the_rows_the_dense_gradient_is_nonzero_on():
    """ODPL_SSD_ROWS_ONLY: same three scalars; the gradient rows of the listed priors equal the dense run's, bit for bit; the list per image is
    ascending and is exactly the set of rows with a non-zero class gradient (cross-entropy: softmax - onehot is never all zero);
    odpl_ssd_level_lists cuts it by source against numpy."""
    import ctypes
    import torch.nn as nn
    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss
    from objectdetectionpl_amd.LightningFunc.utils.SSDUtils import get_dboxes
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(11)
    B, C = 5, 21
    pri = get_dboxes()
    Pn = pri.shape[0]
    off = (torch.randn(B, Pn, 4, generator=g) * 0.5).to(d); cls = torch.randn(B, Pn, C, generator=g).to(d)
    tgt = _ssd_targets(g, B, 3, C)
    tgt = tgt[tgt[:, 0] != 3].to(d)                                              # image 3 has no box: four distinct ids, the fifth image contributes nothing
    crit = SSDLoss(pri, nn.CrossEntropyLoss, "smooth_l1_loss", C, 300)
    g_off, g_cls = torch.empty_like(off), torch.empty_like(cls)
    out_d, _ = crit._launch(off, cls, g_off, g_cls, tgt)
    r_off, r_cls = torch.full_like(off, 7.0), torch.full_like(cls, 7.0)         # rows outside the list must stay untouched
    active = torch.full((B * Pn,), -5, dtype=torch.int32, device=d); cnt = torch.full((B,), -1, dtype=torch.int32, device=d)
    out_r, _ = crit._launch(off, cls, r_off, r_cls, tgt, rows_only=(active, cnt))
    torch.cuda.synchronize()
    assert torch.equal(out_d, out_r)
    nz = (g_cls != 0).any(-1)
    first = [0]
    for cin, hw, a in SSD_SOURCES:
        first.append(first[-1] + hw * hw * a)
    assert first[-1] == Pn
    expect_lists = [[] for _ in SSD_SOURCES]
    for b in range(B):
        n = int(cnt[b])
        rows = active[b * Pn:b * Pn + n].cpu().numpy()
        want = torch.nonzero(nz[b]).flatten().cpu().numpy()
        assert np.array_equal(rows, want), b
        assert torch.equal(r_cls[b, rows], g_cls[b, rows]) and torch.equal(r_off[b, rows], g_off[b, rows])
        mask = torch.ones(Pn, dtype=torch.bool); mask[rows] = False
        assert bool((r_cls[b][mask.to(d)] == 7.0).all()) and bool((r_off[b][mask.to(d)] == 7.0).all())
        for l in range(len(SSD_SOURCES)):
            sel = rows[(rows >= first[l]) & (rows < first[l + 1])]
            expect_lists[l] += [b * (first[l + 1] - first[l]) + int(p) - first[l] for p in sel]
    assert int(cnt[4]) == 0 and int(cnt.sum()) > 0
    lists = torch.full((B * Pn,), -9, dtype=torch.int32, device=d); lc = torch.empty(len(SSD_SOURCES), dtype=torch.int32, device=d)
    arr = (ctypes.c_int32 * (len(first)))(*first)
    _lib.check(_lib.lib().odpl_ssd_level_lists(_lib.ptr(active), _lib.ptr(cnt), B, Pn, len(SSD_SOURCES), arr, _lib.ptr(lists), _lib.ptr(lc),
                                               _lib.current_stream(d)), "odpl_ssd_level_lists")
    torch.cuda.synchronize()
    for l in range(len(SSD_SOURCES)):
        n = int(lc[l])
        assert n == len(expect_lists[l])
        assert lists[B * first[l]:B * first[l] + n].cpu().tolist() == expect_lists[l], l


@pytest.mark.parametrize("cls_name", ["ce_loss", "focal_loss"])
def test_ssd_head_pairs_structured_backward_matches_dense(cls_name):
    """The six cl / bbx pairs on SSD300's source shapes (Cin 512 / 1024 / 512 / 256 / 256 / 256; 4 / 6 priors; 8 732 rows), B = 4: gradients of
    every source map and of all 24 weights / biases through `odpl_retina_heads_backward` per source (positives + mined negatives, ~1 % of
    the rows) == through the dense route (SSDLoss's dense gradient -> aten::convolution_backward of the twelve convolutions).  VERDICT r3
    item 8 / DESIGN section 7."""
    import torch.nn as nn
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss, focal_loss
    from objectdetectionpl_amd.LightningFunc.utils import ssdhead
    from objectdetectionpl_amd.LightningFunc.utils.SSDUtils import get_dboxes
    d = torch.device("cuda:0")
    g = torch.Generator().manual_seed(5)
    B, C = 4, 21
    torch.manual_seed(3)
    cl = [nn.Conv2d(cin, a * C, 3, padding=1) for cin, hw, a in SSD_SOURCES]
    bx = [nn.Conv2d(cin, a * 4, 3, padding=1) for cin, hw, a in SSD_SOURCES]
    for m in cl + bx:
        m.to(d).to(memory_format=torch.channels_last)
    xs0 = [(torch.randn(B, cin, hw, hw, generator=g) * 0.3).to(d).contiguous(memory_format=torch.channels_last) for cin, hw, a in SSD_SOURCES]
    tgt = _ssd_targets(g, B, 4, C).to(d)
    crit = SSDLoss(get_dboxes(), focal_loss if cls_name == "focal_loss" else nn.CrossEntropyLoss, "smooth_l1_loss", C, 300)
    results = {}
    prev_max = SSDLoss.structured_max_boxes_per_image
    SSDLoss.structured_max_boxes_per_image = None                                # whatever the density (the default takes it at <= 1.5 boxes per image)
    for fused in (True, False):
        SSDLoss.fuse_head_backward = fused
        try:
            xs = [x.clone().requires_grad_() for x in xs0]
            for m in cl + bx:
                m.zero_grad(set_to_none=True)
            out_bbx, out_cl = ssdhead.head_pairs(cl, bx, xs, C)
            assert out_bbx.shape == (B, 8732, 4) and out_cl.shape == (B, 8732, C)
            assert (ssdhead.fusable_source(out_bbx, out_cl) is not None)
            out = crit((out_bbx, out_cl), tgt)
            (out["loss"] * 1.5).backward()                                       # an upstream factor: applied on the device by both routes
            results[fused] = ([x.grad.clone() for x in xs], [p.grad.clone() for m in cl + bx for p in (m.weight, m.bias)], out["loss"].item())
        finally:
            SSDLoss.fuse_head_backward = True
    SSDLoss.structured_max_boxes_per_image = prev_max
    assert abs(results[True][2] - results[False][2]) <= 1e-6 * abs(results[False][2])      # two forwards of the twelve convolutions (library kernels)
    for k, (a, b) in enumerate(zip(results[True][0] + results[True][1], results[False][0] + results[False][1])):
        scale = float(b.abs().max())
        err = float((a - b).abs().max())
        if scale == 0:                                                            # a source without an active prior (the 1 x 1 map): zeros on both routes
            assert err == 0, k
            continue
        assert err <= 2e-4 * scale + 1e-9, (k, err, scale)


def test_ssd_model_training_step_uses_the_structured_head_backward():
    """SSD(...) in channels-last on the device with `ssdhead.enabled`: forward hosts the twelve head convolutions, the criterion takes the
    rows-only route, and one training step's parameter gradients match the default route's (plain modules, dense gradient) within the
    convolution libraries' rounding."""
    from types import SimpleNamespace
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss
    from objectdetectionpl_amd.model.SSD import SSD
    d = torch.device("cuda:0")
    args = SimpleNamespace(model_name="SSD", cls_criterion="ce_loss", coord_criterion="smooth_l1_loss", data_module="synthetic", lr=1e-3, weight_decay=1e-5)
    torch.manual_seed(0)
    model = SSD([str(i) for i in range(21)], args).to(d).to(memory_format=torch.channels_last).train()
    g = torch.Generator().manual_seed(2)
    x = torch.rand(2, 3, 300, 300, generator=g).to(d).contiguous(memory_format=torch.channels_last)
    y = _ssd_targets(g, 2, 3, 21).to(d)
    from objectdetectionpl_amd import _lib
    from objectdetectionpl_amd.LightningFunc.utils import ssdhead
    grads, launches = {}, {}
    for fused in (True, False):
        prev = ssdhead.enabled, SSDLoss.structured_max_boxes_per_image
        ssdhead.enabled = fused
        SSDLoss.structured_max_boxes_per_image = None                            # 3 boxes per image here: the default would take the dense route
        try:
            model.zero_grad(set_to_none=True)
            out = model(x)
            assert hasattr(out[0], "_odpl_ssd_src") == fused
            d_ = model.criterion(out, y)
            assert (type(d_["loss"].grad_fn).__name__ == "_SSDLossHeadFnBackward") == fused
            d_["loss"].backward()
            grads[fused] = {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}
        finally:
            ssdhead.enabled, SSDLoss.structured_max_boxes_per_image = prev
    out = model(x)                                                               # defaults: hosted heads, 3 boxes per image -> the dense route
    assert hasattr(out[0], "_odpl_ssd_src") and type(model.criterion(out, y)["loss"].grad_fn).__name__ == "_SSDLossFnBackward"
    assert type(model.criterion(model(x), y[:2])["loss"].grad_fn).__name__ == "_SSDLossHeadFnBackward"          # one box per image: structured
    assert grads[True].keys() == grads[False].keys() and len(grads[True]) > 40
    for n in grads[True]:
        a, b = grads[True][n].flatten().double(), grads[False][n].flatten().double()
        if float(b.norm()) == 0:                                                  # e.g. the box head of a source without a positive
            assert float(a.norm()) == 0, n
            continue
        cos = float((a @ b) / (a.norm() * b.norm() + 1e-30))
        assert cos > 0.9999, (n, cos)


@pytest.mark.parametrize("C,Pn,B,per_image", [(21, 8732, 3, 11), (230, 1500, 2, 3), (6, 17000, 2, 2), (21, 2000, 40, 30)])
def test_ssd_loss_vs_oracle_over_the_kernel_s_routes(C, Pn, B, per_image):
    """[r4] the reworked image kernel against the CPU oracle (autograd of oracle/ssd.py) on shapes that take its different routes:
    21 classes / 8 732 priors / 11 annotations per image (LDS tile, two chunks of eight annotations); 230 classes (no LDS tile: rows
    straight from global memory, bisection for the hard negatives); 17 000 priors (more than a thread holds in registers: the sweeps
    fall back to memory); 40 images x 30 rows = 1 200 target rows (more than one per thread: the image ids' all-against-all loop)."""
    from oracle import ssd as oracle_ssd
    import torch.nn as nn
    from objectdetectionpl_amd.LightningFunc.losses_dense import SSDLoss
    g = torch.Generator().manual_seed(C + Pn)
    pri = torch.cat([torch.rand(Pn, 2, generator=g), torch.rand(Pn, 2, generator=g) * 0.4 + 0.05], 1)
    off = (torch.randn(B, Pn, 4, generator=g) * 0.5).requires_grad_()
    cls = torch.randn(B, Pn, C, generator=g).requires_grad_()
    tgt = _ssd_targets(g, B, per_image, C)
    ref = oracle_ssd.ssd_loss(off, cls, tgt, pri, "ce_loss", "smooth_l1_loss")
    ref["loss"].backward()
    d = torch.device("cuda:0")
    o2, c2 = off.detach().to(d).requires_grad_(), cls.detach().to(d).requires_grad_()
    crit = SSDLoss(pri, nn.CrossEntropyLoss, "smooth_l1_loss", C, 300)
    out = crit((o2, c2), tgt.to(d))
    out["loss"].backward()
    for k in ("loss", "Localization", "Classification"):
        a, b = out[k].item(), ref[k].item()
        assert abs(a - b) <= 1e-4 * abs(b) + 1e-6, (k, a, b)
    for got, want, name in ((o2.grad, off.grad, "grad_off"), (c2.grad, cls.grad, "grad_cls")):
        got, want = got.cpu().numpy(), want.numpy()
        assert np.array_equal(got != 0, want != 0), name                          # matched set / mined negatives: index work
        assert np.allclose(got, want, rtol=2e-4, atol=1e-6 * np.abs(want).max() + 1e-12), (name, np.abs(got - want).max())


def test_ssd_model_under_autocast_keeps_the_plain_head_modules():
    """The hosted head node runs aten::convolution_backward itself: under autocast (bf16 activations, fp32 master weights) the model
    therefore takes the reference's plain modules and the step still trains."""
    from types import SimpleNamespace
    from objectdetectionpl_amd.model.SSD import SSD
    d = torch.device("cuda:0")
    args = SimpleNamespace(model_name="SSD", cls_criterion="ce_loss", coord_criterion="smooth_l1_loss", data_module="synthetic", lr=1e-3, weight_decay=1e-5)
    torch.manual_seed(0)
    model = SSD([str(i) for i in range(21)], args).to(d).to(memory_format=torch.channels_last).train()
    g = torch.Generator().manual_seed(4)
    x = torch.rand(2, 3, 300, 300, generator=g).to(d).contiguous(memory_format=torch.channels_last)
    y = _ssd_targets(g, 2, 1, 21).to(d)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        out = model(x)
    assert not hasattr(out[0], "_odpl_ssd_src") and out[1].dtype == torch.bfloat16
    loss = model.criterion(out, y)["loss"]
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in model.parameters())
