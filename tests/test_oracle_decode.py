import numpy as np
import torch

from oracle import decode

A18 = [12, 16, 19, 36, 40, 28, 36, 75, 76, 55, 72, 146, 142, 110, 192, 243, 459, 401]


def T(a):
    return torch.from_numpy(np.asarray(a))


def test_decode_matches_reference(golden):
    g = golden("decode")
    heads = [T(g[f"head{i}"]) for i in range(3)]
    b, c = decode.decode_heads(heads, 5, A18, [[0, 1, 2], [3, 4, 5], [6, 7, 8]], [8, 16, 32])
    # fp32 class: ATen's sigmoid/exp take different vector paths on strided views vs contiguous copies (1 ulp)
    assert np.allclose(b.numpy(), g["boxes"], rtol=1e-6, atol=1e-7) and np.allclose(c.numpy(), g["confs"], rtol=1e-6, atol=1e-7)
    b, c = decode.decode_scale(T(g["ns_head"]), 3, [1.5, 2.0, 3.0, 1.0], 2, scale_x_y=1.2)
    # sigmoid on a strided view vs a contiguous copy takes different vector paths in ATen: 1 ulp
    assert np.allclose(b.numpy(), g["ns_boxes"], rtol=1e-6, atol=1e-7) and np.allclose(c.numpy(), g["ns_confs"], rtol=1e-6, atol=1e-7)
