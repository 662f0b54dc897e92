"""Mirror of `LightningFunc/utils/SSDUtils.py:5-27`: the SSD300 prior table, built on the device."""
import ctypes
import math

import torch

from ... import _lib


def get_dboxes(smin=0.07, smax=0.9, ars=(1, 2, (1 / 2.0), 3, (1 / 3.0)), fks=(38, 19, 10, 5, 3, 1),
               num_boxes=(3, 5, 5, 5, 3, 3), device="cuda"):
    """-> f32 [8732, 4] (cx, cy, w, h), clamped to <= 1; bit-identical to the reference's table."""
    m = len(fks)
    if m != 6:
        raise ValueError("the SSD prior table has six feature maps")
    sks = [round(smin + (((smax - smin) / (m - 1)) * (k - 1)), 2) for k in range(1, m + 1)]      # SSDUtils.py:7
    fk = (ctypes.c_int32 * 6)(*fks)
    per = (ctypes.c_int32 * 6)(*[n + 1 for n in num_boxes])
    wh = (ctypes.c_float * 72)()
    n = 0
    for k in range(6):
        s_extra = math.sqrt(sks[k] * sks[min(k + 1, m - 1)])                                       # SSDUtils.py:17
        wh[(k * 6) * 2], wh[(k * 6) * 2 + 1] = s_extra, s_extra
        for b, ar in enumerate(ars[:num_boxes[k]], 1):
            wh[(k * 6 + b) * 2], wh[(k * 6 + b) * 2 + 1] = sks[k] * math.sqrt(ar), sks[k] / math.sqrt(ar)   # :22-23
        n += fks[k] * fks[k] * (num_boxes[k] + 1)
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.OdplError("get_dboxes builds the table on the HIP device; pass a cuda device")
    out = torch.empty(n, 4, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().odpl_ssd_priors(fk, per, wh, _lib.ptr(out), n, _lib.current_stream(dev)), "odpl_ssd_priors")
    return out
