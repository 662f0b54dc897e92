// IoU family entry points (accuracy.py:6-114, 297-303; losses.py:375-403) and the SSD prior table
// (SSDUtils.py:5-27).  Elementwise, HBM-bound: one thread per pair, 16-B loads where the layout allows.
#include "odpl_common.hpp"
#include "odpl_boxes_dev.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

__device__ __forceinline__ Box4 load_box(const float* p, long long i, long long n, int layout_t, int xyxy) {
    float a, b, c, d;
    if (layout_t) { a = p[i]; b = p[n + i]; c = p[2 * n + i]; d = p[3 * n + i]; }          // [4,N]
    else { const float4 v = reinterpret_cast<const float4*>(p)[i]; a = v.x; b = v.y; c = v.z; d = v.w; }  // [N,4]
    if (xyxy) { Box4 r; r.x1 = a; r.y1 = b; r.x2 = c; r.y2 = d; return r; }
    return corners_of(a, b, c, d);
}

// kind: 0 bbox_iou (+1), 1..4 bbox_iou_v5 {IoU,GIoU,DIoU,CIoU}, 5 iou (corner, no eps)
__global__ __launch_bounds__(256) void pair_iou_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                       float* __restrict__ out, long long n, long long n1, int kind,
                                                       int xyxy, int layout_t) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Box4 a = load_box(b1, n1 == 1 ? 0 : i, n1, layout_t, xyxy);          // n1 == 1: broadcast box1 (accuracy.py:271 use)
    const Box4 b = load_box(b2, i, n, layout_t, xyxy);
    float v;
    if (kind == 0) v = iou_plus1(a, b);
    else if (kind == 5) v = iou_corner(a, b);
    else v = iou_v5(a, b, kind - 1);
    out[i] = v;
}

__global__ __launch_bounds__(256) void wh_iou_kernel(const float* __restrict__ anchors, const float* __restrict__ gwh,
                                                     float* __restrict__ out, int A, long long nT) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nT * A) return;
    const int a = (int)(i / nT); const long long t = i - (long long)a * nT;
    const float w1 = anchors[a * 2], h1 = anchors[a * 2 + 1], w2 = gwh[t * 2], h2 = gwh[t * 2 + 1];
    const float inter = fminf(w1, w2) * fminf(h1, h2);                          // accuracy.py:297-303
    out[i] = inter / ((w1 * h1 + 1e-16f) + w2 * h2 - inter);
}

__global__ __launch_bounds__(256) void iou_matrix_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                         float* __restrict__ out, long long N, int M, int xywh) {
    // RetinaNetLoss.box_iou (losses.py:375-403): [N,4] x [M,4] -> [N,M]
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N * M) return;
    const long long r = i / M; const int c = (int)(i - r * M);
    const float4 u = reinterpret_cast<const float4*>(b1)[r], v = reinterpret_cast<const float4*>(b2)[c];
    Box4 a, b;
    if (xywh) { a = corners_of(u.x, u.y, u.z, u.w); b = corners_of(v.x, v.y, v.z, v.w); }
    else { a.x1 = u.x; a.y1 = u.y; a.x2 = u.z; a.y2 = u.w; b.x1 = v.x; b.y1 = v.y; b.x2 = v.z; b.y2 = v.w; }
    out[i] = iou_plus1_noeps(a, b);
}

__global__ __launch_bounds__(256) void xywh2xyxy_kernel(const float* __restrict__ in, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 v = reinterpret_cast<const float4*>(in)[i];
    const Box4 b = corners_of(v.x, v.y, v.z, v.w);                               // accuracy.py:289-295
    reinterpret_cast<float4*>(out)[i] = make_float4(b.x1, b.y1, b.x2, b.y2);
}

struct PriorParams { int fk[6], per[6], first[7]; float wh[6][6][2]; };

__global__ __launch_bounds__(256) void ssd_priors_kernel(const PriorParams P, float* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int k = 0;
    while (k < 5 && i >= P.first[k + 1]) ++k;
    const int per = P.per[k], r = i - P.first[k], box = r % per, cell = r / per;
    const int ci = cell / P.fk[k], cj = cell % P.fk[k];                          // itertools.product(i, j): i slowest
    // SSDUtils.py:14-15: python doubles, then .float(), then clamp(max=1) (SSDUtils.py:26-27)
    const double cx = ((double)ci + 0.5) / (double)P.fk[k], cy = ((double)cj + 0.5) / (double)P.fk[k];
    float4 o;
    o.x = fminf((float)cx, 1.0f); o.y = fminf((float)cy, 1.0f);
    o.z = fminf(P.wh[k][box][0], 1.0f); o.w = fminf(P.wh[k][box][1], 1.0f);
    reinterpret_cast<float4*>(out)[i] = o;
}
}  // namespace

extern "C" int odpl_pair_iou(const float* box1, const float* box2, float* out, int64_t n, int64_t n1, int32_t kind,
                             int32_t x1y1x2y2, int32_t layout_4xn, void* stream) {
    if (!box1 || !box2 || !out || n < 0 || kind < 0 || kind > 5 || (n1 != n && n1 != 1)) return ODPL_EINVAL;
    if (!layout_4xn && ((((uintptr_t)box1) | ((uintptr_t)box2)) & 15)) return ODPL_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(pair_iou_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       box1, box2, out, (long long)n, (long long)n1, kind, x1y1x2y2, layout_4xn);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_bbox_wh_iou(const float* anchors, const float* gwh, float* out, int32_t A, int64_t nT, void* stream) {
    if (!anchors || !gwh || !out || A < 1 || nT < 0) return ODPL_EINVAL;
    if (nT == 0) return 0;
    hipLaunchKernelGGL(wh_iou_kernel, dim3((unsigned)((nT * A + 255) / 256)), dim3(256), 0, (hipStream_t)stream, anchors, gwh, out, A, (long long)nT);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_box_iou_matrix(const float* box1, const float* box2, float* out, int64_t N, int32_t M, int32_t xywh, void* stream) {
    if (!box1 || !box2 || !out || N < 0 || M < 0 || ((((uintptr_t)box1) | ((uintptr_t)box2)) & 15)) return ODPL_EINVAL;
    if (N == 0 || M == 0) return 0;
    hipLaunchKernelGGL(iou_matrix_kernel, dim3((unsigned)((N * M + 255) / 256)), dim3(256), 0, (hipStream_t)stream, box1, box2, out, (long long)N, M, xywh);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_xywh2xyxy(const float* in, float* out, int64_t n, void* stream) {
    if (!in || !out || n < 0 || ((((uintptr_t)in) | ((uintptr_t)out)) & 15)) return ODPL_EINVAL;
    if (n == 0) return 0;
    hipLaunchKernelGGL(xywh2xyxy_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, out, (long long)n);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_ssd_priors(const int32_t* host_fk, const int32_t* host_per_cell, const float* host_wh, float* out, int64_t n_out, void* stream) {
    if (!host_fk || !host_per_cell || !host_wh || !out || ((uintptr_t)out & 15)) return ODPL_EINVAL;
    PriorParams P;
    long long tot = 0;
    for (int k = 0; k < 6; ++k) {
        P.fk[k] = host_fk[k]; P.per[k] = host_per_cell[k];
        if (P.fk[k] < 1 || P.per[k] < 1 || P.per[k] > 6) return ODPL_EINVAL;
        P.first[k] = (int)tot; tot += (long long)P.fk[k] * P.fk[k] * P.per[k];
        for (int b = 0; b < 6; ++b) { P.wh[k][b][0] = host_wh[(k * 6 + b) * 2]; P.wh[k][b][1] = host_wh[(k * 6 + b) * 2 + 1]; }
    }
    P.first[6] = (int)tot;
    if (tot != n_out || tot > 0x7fffffffLL) return ODPL_EINVAL;
    hipLaunchKernelGGL(ssd_priors_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, P, out, (int)tot);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
