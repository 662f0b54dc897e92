// Device helpers shared by the YOLO assignment kernels (odpl_yolo.hip, odpl_yolo_targets.hip).
#pragma once
#include "odpl_common.hpp"

namespace odpl {

// target row -> indices, accuracy.py:327-337
struct TDec {
    long long b, lab, gi, gj;
    float gx, gy, gw, gh;
    int best;
    unsigned over;      // bit a: anchor-shape IoU with anchor a > ignore_thres
};

__device__ __forceinline__ float wh_iou_dev(float w1, float h1, float w2, float h2) {
    // accuracy.py:297-303, same association: (w1*h1 + 1e-16) + w2*h2 - inter
    const float inter = fminf(w1, w2) * fminf(h1, h2);
    const float uni = (w1 * h1 + 1e-16f) + w2 * h2 - inter;
    return inter / uni;
}

__device__ __forceinline__ TDec decode_target(const float* __restrict__ y, int G, int A,
                                              const float* aw, const float* ah, float thr) {
    TDec d;
    const float fG = (float)G;
    d.b = trunc_ll(y[0]);
    d.lab = trunc_ll(y[1]);
    d.gx = y[2] * fG; d.gy = y[3] * fG; d.gw = y[4] * fG; d.gh = y[5] * fG;
    d.gi = trunc_ll(d.gx);
    d.gj = trunc_ll(d.gy);
    float best = wh_iou_dev(aw[0], ah[0], d.gw, d.gh);
    d.best = 0;
    d.over = best > thr ? 1u : 0u;
    for (int a = 1; a < A; ++a) {
        const float v = wh_iou_dev(aw[a], ah[a], d.gw, d.gh);
        if (v > thr) d.over |= 1u << a;
        // torch.max(0): first index on ties; a NaN wins and then sticks
        if (best == best && (v > best || v != v)) { best = v; d.best = a; }
    }
    return d;
}

// python negative-index wrap; returns false when the reference would raise IndexError
__device__ __forceinline__ bool wrap_idx(long long& i, int size) {
    if (i < 0) i += size;
    return i >= 0 && i < size;
}


}  // namespace odpl
