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
    assert _lib.lib().odpl_abi_version() == _lib.ABI_VERSION == 2


def test_struct_layout_matches_header():
    from objectdetectionpl_amd import _lib
    # odpl_yolo_scale_t: 2 pointers + 2 int32 + 2*8 floats
    assert ctypes.sizeof(_lib.YoloScale) == 8 + 8 + 4 + 4 + 4 * 16
    # odpl_yolo_loss_desc_t: 8 int32 + 3 float (+pad) + pointer + 4 scales
    assert _lib.YoloLossDesc.target.offset == 48
    assert ctypes.sizeof(_lib.YoloLossDesc) == 56 + 4 * ctypes.sizeof(_lib.YoloScale)


def test_workspace_query_is_host_only():
    from objectdetectionpl_amd import _lib
    d = _lib.YoloLossDesc()
    d.B, d.C, d.nT, d.nscales = 16, 80, 112, 3
    for i, G in enumerate((52, 26, 13)):
        d.scale[i].G, d.scale[i].A, d.scale[i].head = G, 3, 0x1000
    n = _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d))
    assert 16 * 3 * (52 * 52 + 26 * 26 + 13 * 13) * 4 < n < 4 << 20
    d.nscales = 9
    assert _lib.lib().odpl_yolo_region_loss_workspace_bytes(ctypes.byref(d)) == 0


def test_bad_arguments_return_einval_without_touching_the_gpu():
    from objectdetectionpl_amd import _lib
    L = _lib.lib()
    d = _lib.YoloLossDesc()
    assert L.odpl_yolo_region_loss(ctypes.byref(d), None, 0, None, None, None, None) == -1
    assert L.odpl_scale_inplace_if_not_one(None, 4, None, None) == -1
    assert L.odpl_profile_enable(99, 1) == -1
    assert L.odpl_tune_set(99, 1) == -1
