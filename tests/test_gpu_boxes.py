"""GPU parity: IoU family and prior tables (C ABI) vs vectors captured from the reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def T(a):
    return torch.from_numpy(np.asarray(a)).cuda()


def test_iou_family_vs_golden(golden):
    from objectdetectionpl_amd.LightningFunc import accuracy as A
    g = golden("iou")
    a, b, ax, bx = (T(g[k]) for k in ("a_xywh", "b_xywh", "a_xyxy", "b_xyxy"))
    assert np.array_equal(A.bbox_iou(ax, bx).cpu().numpy(), g["bbox_iou_xyxy"])
    assert np.array_equal(A.bbox_iou(a, b, x1y1x2y2=False).cpu().numpy(), g["bbox_iou_xywh"])
    assert np.array_equal(A.bbox_iou(ax[0:1], bx).cpu().numpy(), g["bbox_iou_bcast"])
    for mode, kw in (("iou", {}), ("giou", dict(GIoU=True)), ("diou", dict(DIoU=True))):
        for lay, u, v, xy in (("xywh", a, b, False), ("xyxy", ax, bx, True)):
            got = A.bbox_iou_v5(u.t(), v.t(), x1y1x2y2=xy, **kw).cpu().numpy()
            assert np.array_equal(got, g[f"v5_{mode}_{lay}"], equal_nan=True), (mode, lay)
    for lay, u, v, xy in (("xywh", a, b, False), ("xyxy", ax, bx, True)):     # atan: device libm, f32 tolerance
        got = A.bbox_iou_v5(u.t(), v.t(), x1y1x2y2=xy, CIoU=True).cpu().numpy()
        assert np.allclose(got, g[f"v5_ciou_{lay}"], rtol=1e-5, atol=1e-6, equal_nan=True)
    wh = torch.stack([A.bbox_wh_iou(an, T(g["wh_gwh"])) for an in T(g["wh_anchors"])])
    assert np.array_equal(wh.cpu().numpy(), g["wh_iou"])
    assert np.array_equal(A.iou(T(g["corner_a"]), T(g["corner_b"])).cpu().numpy(), g["corner_iou"], equal_nan=True)
    assert np.array_equal(A.box_iou_matrix(a[:64], b[:16], "xywh").cpu().numpy(), g["retina_box_iou_xywh"])
    assert np.array_equal(A.box_iou_matrix(ax[:64], bx[:16], "xyxy").cpu().numpy(), g["retina_box_iou_xyxy"])
    assert np.array_equal(A.xywh2xyxy(a).cpu().numpy(), g["a_xyxy"])


def test_ssd_priors_bit_exact(golden):
    from objectdetectionpl_amd.LightningFunc.utils.SSDUtils import get_dboxes
    g = golden("anchors")
    t = get_dboxes().cpu().numpy()
    assert t.shape == (8732, 4) and np.array_equal(t, g["ssd_dboxes"])


@pytest.mark.parametrize("cin,cout,h", [(64, 64, 26), (256, 128, 13), (128, 256, 9), (32, 64, 20)])
def test_conv1x1_as_fp32_mfma_gemm_vs_conv2d(cin, cout, h):
    """csrc/odpl_gemm1x1.hip (a probe, not used by the models: it ties MIOpen on the network's 1x1 layers): a 1x1
    convolution on channels-last tensors as a GEMM on v_mfma_f32_32x32x2_f32 -- forward and backward-data against F.conv2d, both tiles,
    a row count that is not a multiple of the tile."""
    import torch.nn.functional as F
    from objectdetectionpl_amd import _lib
    L = _lib.lib()
    g = torch.Generator().manual_seed(cin + cout)
    B = 3
    x = torch.randn(B, cin, h, h, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    w = (torch.randn(cout, cin, 1, 1, generator=g) * 0.1).cuda().contiguous(memory_format=torch.channels_last)
    y_ref = F.conv2d(x, w)
    gy = torch.randn(y_ref.shape, generator=g).cuda().contiguous(memory_format=torch.channels_last)
    gx_ref = torch.ops.aten.convolution_backward(gy, x, w, None, [1, 1], [0, 0], [1, 1], False, [0, 0], 1, [True, False, False])[0]
    M, st = B * h * h, _lib.current_stream(x.device)
    for tile in (64, 128):
        if cout % tile == 0:
            y = torch.empty_like(y_ref)
            _lib.check(L.odpl_conv1x1_forward(_lib.ptr(x), _lib.ptr(w), _lib.ptr(y), M, cin, cout, tile, st), "fwd")
            assert torch.allclose(y, y_ref, rtol=1e-4, atol=1e-5 * float(y_ref.abs().max()))
        if cin % tile == 0:
            gx = torch.empty_like(x)
            _lib.check(L.odpl_conv1x1_backward_data(_lib.ptr(gy), _lib.ptr(w), _lib.ptr(gx), M, cin, cout, tile, st), "bwd")
            assert torch.allclose(gx, gx_ref, rtol=1e-4, atol=1e-5 * float(gx_ref.abs().max()))
