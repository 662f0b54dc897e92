"""Mirror of `LightningFunc/utils/RetinaUtils.py:6-80`: the RetinaNet anchor table, built on the device."""
import ctypes
import math

import torch

from ... import _lib


def get_anchor_wh():
    """RetinaUtils.py:62-80: python-double maths, one rounding to f32.  -> [5][9][2] nested list."""
    out = []
    for s in (32 * 32., 64 * 64., 128 * 128., 256 * 256., 512 * 512.):
        lvl = []
        for ar in (1 / 2., 1 / 1., 2 / 1.):
            h = math.sqrt(s / ar)
            w = ar * h
            for sr in (1., pow(2, 1 / 3.), pow(2, 2 / 3.)):
                lvl.append((w * sr, h * sr))
        out.append(lvl)
    return out


def get_anchor_boxes(input_size, device="cuda"):
    """RetinaUtils.py:6-31.  input_size: tensor / sequence (w, h).  -> f32 [sum fm_h*fm_w*9, 4] (cx,cy,w,h)
    on `device`, bit-identical to the reference's table."""
    size = torch.as_tensor(input_size, dtype=torch.float32).cpu()
    fm = (ctypes.c_int32 * 10)()
    grid = (ctypes.c_float * 10)()
    n = 0
    for i in range(5):
        f = (size / pow(2., i + 3)).ceil()          # f32 tensor maths, as the reference (RetinaUtils.py:19)
        g = size / f                                 # RetinaUtils.py:24
        fm[2 * i], fm[2 * i + 1] = int(f[0]), int(f[1])
        grid[2 * i], grid[2 * i + 1] = float(g[0]), float(g[1])
        n += int(f[0]) * int(f[1]) * 9
    wh = (ctypes.c_float * 90)()
    for i, lvl in enumerate(get_anchor_wh()):
        for k, (w, h) in enumerate(lvl):
            wh[(i * 9 + k) * 2], wh[(i * 9 + k) * 2 + 1] = w, h
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.OdplError("get_anchor_boxes builds the table on the HIP device; pass a cuda device")
    out = torch.empty(n, 4, dtype=torch.float32, device=dev)
    _lib.check(_lib.lib().odpl_retina_anchors(fm, grid, wh, _lib.ptr(out), n, _lib.current_stream(dev)), "odpl_retina_anchors")
    return out
