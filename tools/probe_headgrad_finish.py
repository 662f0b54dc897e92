#!/usr/bin/env python3
"""Times `headgrad_stream` / `headgrad_finish` (odpl_head_conv_backward_multi) at BASELINE config 5's geometry -- three bf16 heads,
B = 32, G = 80 / 40 / 20, Cin = 256 / 512 / 1024, A = 3, no = 85, ~1 300 matched rows per head (interleaved cells) -- with parts of
the work switched off, to see which part of the finish launch the time is in.  One JSON line per variant."""
import ctypes
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from objectdetectionpl_amd import _lib

DEV = torch.device("cuda:0")


def setup(B=32, rows=1300, bf16=True, Cins=(256, 512, 1024), Gs=(80, 40, 20), A=3, no=85, seed=0, v5_targets=0):
    g = torch.Generator().manual_seed(seed)
    keep, heads, grads = [], [], []
    for Cin, G in zip(Cins, Gs):
        G2 = G * G
        N = B * G2
        x = torch.randn(N, Cin, generator=g).to(DEV)
        x = x.bfloat16() if bf16 else x
        w = (torch.randn(A * no, Cin, generator=g) * 0.1).to(DEV)
        gd = torch.randn(N * A, generator=g).to(DEV)
        gr = torch.randn(max(rows, 1), no, generator=g).to(DEV)
        if v5_targets:
            # YOLOv5's row list (losses.py:159-315 build_targets): every target passes the ratio test on ~2 of 3 anchors; a passing
            # (target, anchor) gives the centre cell and two neighbours; rows in offset-major, anchor-major, target order; the list
            # has the CAPACITY 5 * A * nT, rows beyond the produced count carry cell -1.  Planar cell = (b*A + a)*G2 + s.
            nT = v5_targets
            tb = torch.arange(nT) * B // nT
            ty = torch.randint(1, G - 1, (nT,), generator=g); tx = torch.randint(1, G - 1, (nT,), generator=g)
            ok = torch.rand(A, nT, generator=g) < 0.65
            dy = torch.where(torch.rand(nT, generator=g) < 0.5, -1, 1); dx = torch.where(torch.rand(nT, generator=g) < 0.5, -1, 1)
            lst = []
            for oy, ox in ((0, 0), (None, 0), (0, None)):
                for a in range(A):
                    m = ok[a]
                    yy = ty + (dy if oy is None else 0); xx = tx + (dx if ox is None else 0)
                    lst.append(((tb * A + a) * G2 + yy * G + xx)[m])
            cells = torch.cat(lst).int()
            n_dev = torch.tensor([cells.numel()], dtype=torch.int32, device=DEV); keep.append(n_dev)
            rows = 5 * A * nT
            cells = torch.cat([cells, torch.full((rows - cells.numel(),), -1, dtype=torch.int32)]).to(DEV)
            gr = torch.randn(rows, no, generator=g).to(DEV)
            layout = 0
        else:
            # anchor-major blocks of random cells, some cells twice; interleaved layout: cell = n * A + a
            cell_n = torch.randint(0, N, (max(rows, 1),), generator=g)
            cell_a = (torch.arange(max(rows, 1)) * A // max(rows, 1)).clamp(max=A - 1)
            cells = (cell_n * A + cell_a).int().to(DEV)
            layout = 1
        gx, gw, gb = torch.empty_like(x), torch.empty_like(w), torch.empty(A * no, device=DEV)
        keep += [x, w, gd, gr, cells, gx, gw, gb]
        heads.append(dict(x=x, w=w, gx=gx, gw=gw, gb=gb, Cin=Cin))
        grads.append(_lib.SparseGrad(dense=gd.data_ptr(), rows=gr.data_ptr(), row_cell=cells.data_ptr(), n_rows_dev=n_dev.data_ptr() if v5_targets else None,
                                     n_rows=rows, B=B, G2=G2, A=A, no=no,
                                     dense_ch=4, layout=layout))
    return keep, heads, grads


def run(tag, rows=1300, want_x=True, want_w=True, bf16=True, steps=20, **kw):
    keep, heads, grads = setup(rows=rows, bf16=bf16, **kw)
    n = len(heads)
    H = (_lib.HeadConv * n)(*[_lib.HeadConv(x=h["x"].data_ptr(), w=h["w"].data_ptr(), grad_x=h["gx"].data_ptr() if want_x else None,
                                            grad_w=h["gw"].data_ptr() if want_w else None, grad_b=h["gb"].data_ptr() if want_w else None,
                                            upstream=None, Cin=h["Cin"], bf16=1 if bf16 else 0) for h in heads])
    S = (_lib.SparseGrad * n)(*grads)
    L = _lib.lib()
    nb = L.odpl_head_conv_backward_multi_workspace_bytes(n, H, S)
    assert nb > 0
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = _lib.current_stream(DEV)
    for i in range(steps + 3):
        if i == 3:
            _lib.profile_enable("headgrad_stream", True); _lib.profile_enable("headgrad_finish", True)
        _lib.check(L.odpl_head_conv_backward_multi(n, H, S, _lib.ptr(ws), nb, st), "multi")
    torch.cuda.synchronize()
    out = {"variant": tag, "rows_per_head": rows, "grad_x": want_x, "grad_w": want_w, "bf16": bf16}
    for k in ("headgrad_stream", "headgrad_finish"):
        _lib.profile_enable(k, False)
        ms = _lib.profile_collect(k)[0]
        out[k + "_us"] = round(1e3 * sum(ms) / max(len(ms), 1), 2)
    print(json.dumps(out), flush=True)


def trace(rows=1300, **kw):
    """Per-phase wall-clock stamps of the finish launch's workgroups (a library built with -DODPL_FINISH_TRACE)."""
    import numpy as np
    L = _lib.lib()
    L.odpl_head_conv_trace.argtypes = [ctypes.c_void_p]
    buf = torch.zeros(65536 * 8, dtype=torch.int64, device=DEV)
    keep, heads, grads = setup(rows=rows, **kw)
    n = len(heads)
    bf16 = kw.get("bf16", True)
    H = (_lib.HeadConv * n)(*[_lib.HeadConv(x=h["x"].data_ptr(), w=h["w"].data_ptr(), grad_x=h["gx"].data_ptr(), grad_w=h["gw"].data_ptr(),
                                            grad_b=h["gb"].data_ptr(), upstream=None, Cin=h["Cin"], bf16=1 if bf16 else 0) for h in heads])
    S = (_lib.SparseGrad * n)(*grads)
    nb = L.odpl_head_conv_backward_multi_workspace_bytes(n, H, S)
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    st = _lib.current_stream(DEV)
    for _ in range(3):
        _lib.check(L.odpl_head_conv_backward_multi(n, H, S, _lib.ptr(ws), nb, st), "multi")
    torch.cuda.synchronize()
    assert L.odpl_head_conv_trace(buf.data_ptr()) == 0
    _lib.check(L.odpl_head_conv_backward_multi(n, H, S, _lib.ptr(ws), nb, st), "multi")
    torch.cuda.synchronize()
    L.odpl_head_conv_trace(None)
    t = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    live = t[:, 0] > 0
    t = t[live]
    t0 = t[:, 0].min()
    t = t[t[:, 7] > 0]                                     # workgroups that left early (dropped rows, no owner) carry no end stamp
    kind = np.where(t[:, 4] > 0, "c", np.where(t[:, 2] > 0, "b", "a"))
    print(json.dumps({"blocks": int(live.sum()), "span_us": round((t[:, 7].max() - t0) / 100, 2)}))
    for k in "abc":
        m = kind == k
        if not m.any():
            continue
        r = t[m]
        d = {"part": k, "blocks": int(m.sum()), "first_start_us": round((r[:, 0].min() - t0) / 100, 2), "last_start_us": round((r[:, 0].max() - t0) / 100, 2),
             "last_end_us": round((r[:, 7].max() - t0) / 100, 2), "mean_block_us": round(float((r[:, 7] - r[:, 0]).mean()) / 100, 2),
             "max_block_us": round(float((r[:, 7] - r[:, 0]).max()) / 100, 2)}
        if k != "a":
            if k == "c":
                d["start_to_tile_cells_us"] = round(float((r[:, 5] - r[:, 0]).mean()) / 100, 2)
                d["tile_cells_to_dense_part_us"] = round(float((r[:, 6] - r[:, 5]).mean()) / 100, 2)
                d["dense_part_to_compacted_us"] = round(float((r[:, 1] - r[:, 6]).mean()) / 100, 2)
            names = {"b": ["compact", "stage_g", "product"], "c": ["compact", "slots", "matrix", "product"]}[k]
            prev = r[:, 0]
            for i, nm in enumerate(names):
                d[nm + "_us"] = round(float((r[:, i + 1] - prev).mean()) / 100, 2)
                prev = r[:, i + 1]
            d["tail_us"] = round(float((r[:, 7] - prev).mean()) / 100, 2)
        print(json.dumps(d), flush=True)


if __name__ == "__main__":
    if "--trace" in sys.argv:
        trace(v5_targets=224)
        trace()
        trace(rows=112, bf16=False, B=16, Cins=(1024, 512, 256), Gs=(13, 26, 52))
        sys.exit(0)
    run("config 5 row list (224 targets, capacity 3360)", v5_targets=224)
    run("all")
    run("no rows: dense-channel reduce only", rows=0)
    run("dW/db only", want_x=False)
    run("dX only", want_w=False)
    run("all, 300 rows", rows=300)
    run("all, 2600 rows", rows=2600)
    run("config 2 geometry (fp32, B=16, 13/26/52, 112 rows)", rows=112, bf16=False, B=16, Cins=(1024, 512, 256), Gs=(13, 26, 52))
