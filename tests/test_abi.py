"""C-ABI checks that need no GPU: the library loads, exports every symbol include/odpl_hip.h declares,
the ctypes table covers exactly those symbols, and struct layouts match the header."""
import ctypes
import os
import re

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(REPO, "include", "odpl_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|size_t)\s+(odpl_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported_and_bound():
    from objectdetectionpl_amd import _lib
    names = header_functions()
    assert len(names) >= 6
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in odpl_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == names, (set(names) ^ set(_lib.SIGNATURES))
    assert _lib.lib().odpl_abi_version() == _lib.ABI_VERSION == 7


def test_struct_layout_matches_header():
    from objectdetectionpl_amd import _lib
    # odpl_yolo_scale_t: 2 pointers + 2 int32 + 2*8 floats + 4 int64 strides
    assert ctypes.sizeof(_lib.YoloScale) == 8 + 8 + 4 + 4 + 4 * 16 + 8 * 4
    assert _lib.YoloScale.stride.offset == 88
    # odpl_sparse_grad_t: 3 pointers + 7 int32 (+pad)
    assert ctypes.sizeof(_lib.SparseGrad) == 64
    # odpl_yolo_loss_desc_t: 8 int32 + 3 float (+pad) + pointer + 4 scales
    assert _lib.YoloLossDesc.target.offset == 48
    assert ctypes.sizeof(_lib.YoloLossDesc) == 56 + 4 * ctypes.sizeof(_lib.YoloScale)
    # odpl_retina_loss_desc_t: 6 pointers + 6 int32 + 2 floats + 2 pointers (ABI 4)
    assert ctypes.sizeof(_lib.RetinaLossDesc) == 96 and _lib.RetinaLossDesc.pos_anchor.offset == 80
    # odpl_retina_head_t: 5 + 5 + 4 pointers, 4 int64 strides, int32 (+pad); odpl_retina_heads_desc_t: 14 int32 + 3 pointers
    assert ctypes.sizeof(_lib.RetinaHead) == 152 and _lib.RetinaHead.w_stride.offset == 112
    assert ctypes.sizeof(_lib.RetinaHeadsDesc) == 88 and _lib.RetinaHeadsDesc.pos_anchor.offset == 56 and _lib.RetinaHeadsDesc.flags.offset == 84


def test_ctypes_layouts_against_the_compiled_header(tmp_path):
    """sizeof / offsetof as gcc sees include/odpl_hip.h, against the ctypes mirrors (last field of each struct pins the whole layout)"""
    import subprocess
    from objectdetectionpl_amd import _lib
    cases = [("odpl_retina_loss_desc_t", _lib.RetinaLossDesc, "num_pos"), ("odpl_retina_head_t", _lib.RetinaHead, "no"),
             ("odpl_retina_heads_desc_t", _lib.RetinaHeadsDesc, "flags"), ("odpl_ssd_loss_desc_t", _lib.SsdLossDesc, "active_count"), ("odpl_sparse_conv3x3_t", _lib.SparseConv3x3, "stats"),
             ("odpl_pixel_support_t", _lib.PixelSupport, "counts"), ("odpl_sparse_conv3x3_fwd_t", _lib.SparseConv3x3Fwd, "ksize"),
             ("odpl_yolo_scale_t", _lib.YoloScale, "stride"), ("odpl_sparse_grad_t", _lib.SparseGrad, "layout"),
             ("odpl_adam_desc_t", _lib.AdamDesc, "weight_decay"), ("odpl_yolo_nms_desc_t", _lib.YoloNmsDesc, "flags")]
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{REPO}/include/odpl_hip.h"', 'int main(void) {']
    for cname, _, last in cases:
        off = f"offsetof({cname}, {last})" if last else "(size_t)0"
        lines.append(f'  printf("%zu %zu\\n", sizeof({cname}), {off});')
    lines += ['  return 0;', '}']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split("\n")
    for (cname, ct, last), line in zip(cases, out):
        size, off = (int(v) for v in line.split())
        assert ctypes.sizeof(ct) == size, cname
        if last:
            assert getattr(ct, last).offset == off, (cname, last)


def test_workspace_query_is_host_only():
    from objectdetectionpl_amd import _lib
    d = _lib.YoloLossDesc()
    d.B, d.C, d.nT, d.nscales = 16, 80, 112, 3
    for i, G in enumerate((52, 26, 13)):
        d.scale[i].G, d.scale[i].A, d.scale[i].head = G, 3, 0x1000
    n = _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d))
    assert 16 * 3 * (52 * 52 + 26 * 26 + 13 * 13) * 4 < n < 4 << 20
    # the compact (structured) gradient: one float per cell + nT rows of 5+C + nT cells, per scale
    c = _lib.lib().odpl_yolo_region_loss_compact_bytes(ctypes.byref(d))
    assert 16 * 3 * (52 * 52 + 26 * 26 + 13 * 13) * 4 + 3 * 112 * 85 * 4 <= c < 1 << 20
    # with a gradient requested the one-call entry keeps the compact form in its workspace
    d.scale[0].grad = 0x2000
    assert _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d)) == n + c
    # channels-last strides are accepted, arbitrary ones are not
    for i, G in enumerate((52, 26, 13)):
        for k, v in enumerate((255 * G * G, 1, 255 * G, 255)):
            d.scale[i].stride[k] = v
    assert _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d)) > 0
    d.scale[1].stride[3] = 7
    assert _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d)) == 0
    d.scale[1].stride[3] = 255
    d.nscales = 9
    assert _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d)) == 0


def test_bad_arguments_return_einval_without_touching_the_gpu():
    from objectdetectionpl_amd import _lib
    L = _lib.lib()
    d = _lib.YoloLossDesc()
    assert L.odpl_yolo_region_loss(ctypes.byref(d), None, 0, None, None, None, None) == -1
    assert L.odpl_yolo_region_loss_forward(ctypes.byref(d), None, 0, None, 0, None, None, None, None) == -1
    assert L.odpl_yolo_region_loss_backward(ctypes.byref(d), None, 0, None, None) == -1
    assert L.odpl_yolo_compact_view(ctypes.byref(d), None, 0, None) == -1
    assert L.odpl_scale_inplace_if_not_one(None, 4, None, None) == -1
    assert L.odpl_profile_enable(99, 1) == -1
    assert L.odpl_tune_set(99, 1) == -1
