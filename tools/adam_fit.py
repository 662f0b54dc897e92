"""Which roundings does ATen's fused Adam kernel perform?  Exact-rational emulation of candidate forms of each expression against
the states dumped by tools/adam_dump.py (gpurun_out/adam_dump.pt).  Prints the mismatch count of every candidate."""
import math
from fractions import Fraction as F

import numpy as np
import torch

d = {k: v.numpy() for k, v in torch.load("gpurun_out/adam_dump.pt", weights_only=True).items()}
f32 = np.float32
lr, b1, b2, eps, wd = 1e-3, 0.9, 0.999, 1e-8, 1e-2
N = 20000


def rd(x):
    return float(x)          # Fraction -> nearest double


def rf(x):
    return float(f32(x))     # double -> nearest float (as a double)


def step_k(k):
    p, g = d[f"p{k}"][:N], d[f"g{k}"][:N]
    m0 = d[f"m{k}"][:N] if k else np.zeros(N, f32)
    v0 = d[f"v{k}"][:N] if k else np.zeros(N, f32)
    return p, g, m0, v0, d[f"m{k+1}"][:N], d[f"v{k+1}"][:N], d[f"p{k+1}"][:N]


for k in (1, 2):
    p, g, m0, v0, m1, v1, p1 = step_k(k)
    step = k + 1
    bc1 = 1 - math.pow(b1, step); bc2 = 1 - math.pow(b2, step)
    bc1f, bc2sf = rf(bc1), rf(math.sqrt(bc2))
    cnt = {}
    def bump(name, ok):
        cnt[name] = cnt.get(name, 0) + (0 if ok else 1)
    for i in range(N):
        P, G, M, V = float(p[i]), float(g[i]), float(m0[i]), float(v0[i])
        # grad += param * wd
        grads = {"g fma(p,wd,g)": rf(rd(F(P) * F(wd) + F(G))), "g plain": rf(rd(F(G) + F(rd(F(P) * F(wd)))))}
        for gn, gg in grads.items():
            # exp_avg
            ea = {"m fma(b1,m,r(w*g))": rf(rd(F(b1) * F(M) + F(rd(F(1 - b1) * F(gg))))),
                  "m fma(w,g,r(b1*m))": rf(rd(F(1 - b1) * F(gg) + F(rd(F(b1) * F(M)))))}
            for en, e in ea.items():
                bump(f"{gn} | {en}", e == float(m1[i]))
            t1 = rd(F(1 - b2) * F(gg))
            va = {"v fma(b2,v,r(t1*g))": rf(rd(F(b2) * F(V) + F(rd(F(t1) * F(gg))))),
                  "v fma(t1,g,r(b2*v))": rf(rd(F(t1) * F(gg) + F(rd(F(b2) * F(V))))),
                  "v plain": rf(rd(F(rd(F(b2) * F(V))) + F(rd(F(t1) * F(gg)))))}
            for vn, v in va.items():
                bump(f"{gn} | {vn}", v == float(v1[i]))
        # param update from torch's own m1, v1 (isolates this expression)
        M1, V1 = float(m1[i]), float(v1[i])
        ss = rf(lr / bc1f)                                   # double / float->double, to float
        sq = rf(math.sqrt(V1))                               # sqrtf
        den_a = rf(rf(sq / bc2sf) + eps)                     # float division, then + eps in double, to float
        den_b = rf(rd(F(sq) / F(bc2sf)) + eps)               # division in double
        for dn, den in (("den f32 div", den_a), ("den f64 div", den_b)):
            upd = rf(rf(ss * M1) / den)
            bump(f"p -= (ss*m)/den, {dn}", rf(P - upd) == float(p1[i]))
            upd2 = rf(ss * rf(M1 / den))
            bump(f"p -= ss*(m/den), {dn}", rf(P - upd2) == float(p1[i]))
    print(f"step {step}:")
    for n_, c in sorted(cnt.items(), key=lambda kv: kv[1]):
        print(f"  {c:6d}  {n_}")
