// Drop-in for accuracy.build_targets (accuracy.py:305-380): the reference's ten dense outputs,
// produced from the same assignment logic as the fused loss (odpl_yolo.hip).  Used by
// get_yolo_statistics (accuracy.py:382-470) and by the parity tests that compare masks bit for bit.
#include "odpl_common.hpp"
#include "odpl_yolo_dev.hpp"
#include "odpl_boxes_dev.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kThreads = 1024;

struct BTParams {
    const float* pred_boxes; const float* pred_cls; const float* target;
    int B, A, G, C, nT;
    float thr;
    float aw[ODPL_MAX_ANCHORS], ah[ODPL_MAX_ANCHORS];
    int* state;
    float* iou_scores; float* class_mask; uint8_t* obj_mask; uint8_t* noobj_mask;
    float* tx; float* ty; float* tw; float* th; float* tcls; float* tconf;
    int* status;
};

__global__ __launch_bounds__(kThreads) void build_targets_kernel(BTParams P, const float* __restrict__ anchors) {
    const int tid = threadIdx.x, nT = P.nT, G = P.G, A = P.A, B = P.B, C = P.C;
    __shared__ float s_aw[ODPL_MAX_ANCHORS], s_ah[ODPL_MAX_ANCHORS];
    if (tid < A) { s_aw[tid] = anchors[tid * 2]; s_ah[tid] = anchors[tid * 2 + 1]; }
    __syncthreads();
    int f1 = 0, f2 = 0, bad = 0;
    for (int t = tid; t < nT; t += kThreads) {
        const TDec d = decode_target(P.target + (size_t)t * 6, G, A, s_aw, s_ah, P.thr);
        f1 |= (d.b >= B) | (d.best >= A) | (d.gj >= G) | (d.gi >= G);
        f2 |= (d.lab >= C);
        bad |= (d.b < -B) | (d.gj < -G) | (d.gi < -G) | (d.lab < -C);
    }
    const int g1 = __syncthreads_or(f1);
    const int g2 = __syncthreads_or(f2) | g1;
    const int st = __syncthreads_or(bad);
    for (int t = tid; t < nT; t += kThreads) {
        const TDec d = decode_target(P.target + (size_t)t * 6, G, A, s_aw, s_ah, P.thr);
        long long b = d.b, gj = d.gj, gi = d.gi;
        const bool upper_ok = (d.b < B) & (d.gj < G) & (d.gi < G);
        if (!(wrap_idx(b, B) & wrap_idx(gj, G) & wrap_idx(gi, G))) continue;
        if (!g1) {
            const int cell = (((int)b * A + d.best) * G + (int)gj) * G + (int)gi;
            atomicMax(&P.state[cell], t + 1);
            P.obj_mask[cell] = 1; P.noobj_mask[cell] = 0; P.tconf[cell] = 1.0f;    // accuracy.py:345-346, 379
        }
        if (upper_ok)
            for (int a = 0; a < A; ++a) if ((d.over >> a) & 1u)
                P.noobj_mask[(((int)b * A + a) * G + (int)gj) * G + (int)gi] = 0;  // accuracy.py:358
    }
    __syncthreads();
    if (!g2) {
        for (int t = tid; t < nT; t += kThreads) {
            const TDec d = decode_target(P.target + (size_t)t * 6, G, A, s_aw, s_ah, P.thr);
            long long b = d.b, gj = d.gj, gi = d.gi, lab = d.lab;
            if (!(wrap_idx(b, B) & wrap_idx(gj, G) & wrap_idx(gi, G))) continue;
            const int cell = (((int)b * A + d.best) * G + (int)gj) * G + (int)gi;
            const long long raw_lab = d.lab;
            if (wrap_idx(lab, C)) P.tcls[(size_t)cell * C + (int)lab] = 1.0f;       // accuracy.py:374
            const int w = __hip_atomic_load(&P.state[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
            if (w != t) continue;                                                   // last writer wins
            P.tx[cell] = d.gx - floorf(d.gx);                                       // accuracy.py:368-372
            P.ty[cell] = d.gy - floorf(d.gy);
            P.tw[cell] = logf(d.gw / s_aw[d.best] + 1e-16f);
            P.th[cell] = logf(d.gh / s_ah[d.best] + 1e-16f);
            const float* pc = P.pred_cls + (size_t)cell * C;                        // accuracy.py:376
            float best = pc[0]; int bi = 0;
            for (int k = 1; k < C; ++k) { const float v = pc[k]; if (best == best && (v > best || v != v)) { best = v; bi = k; } }
            P.class_mask[cell] = ((long long)bi == raw_lab) ? 1.0f : 0.0f;
            const float* pb = P.pred_boxes + (size_t)cell * 4;                      // accuracy.py:377
            P.iou_scores[cell] = iou_plus1(corners_of(pb[0], pb[1], pb[2], pb[3]), corners_of(d.gx, d.gy, d.gw, d.gh));
        }
    }
    if (tid == 0 && P.status) {
        P.status[0] = P.status[1] = 0;
        P.status[2] = g1 ? 1 : 0; P.status[3] = g2 ? 1 : 0; P.status[4] = st ? 1 : 0;
        P.status[5] = P.status[6] = P.status[7] = 0;
    }
}
}  // namespace

extern "C" int odpl_yolo_build_targets(const float* pred_boxes, const float* pred_cls, const float* target,
                                       const float* anchors, int32_t B, int32_t A, int32_t G, int32_t C, int32_t nT,
                                       float ignore_thres, int32_t* state,
                                       float* iou_scores, float* class_mask, uint8_t* obj_mask, uint8_t* noobj_mask,
                                       float* tx, float* ty, float* tw, float* th, float* tcls, float* tconf,
                                       int32_t* status, void* stream_) {
    if (B < 1 || A < 1 || A > ODPL_MAX_ANCHORS || G < 1 || C < 1 || nT < 0) return ODPL_EINVAL;
    if (!pred_boxes || !pred_cls || !anchors || !state || !iou_scores || !class_mask || !obj_mask || !noobj_mask ||
        !tx || !ty || !tw || !th || !tcls || !tconf || (nT > 0 && !target))
        return ODPL_EINVAL;
    if ((long long)B * A * G * G * C > 0x7fffffffLL) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const size_t cells = (size_t)B * A * G * G;
    ODPL_HIP_TRY(odpl::fill_bytes(state, 0xFF, cells * sizeof(int), stream));
    ODPL_HIP_TRY(odpl::fill_bytes(obj_mask, 0, cells, stream));
    ODPL_HIP_TRY(odpl::fill_bytes(noobj_mask, 1, cells, stream));
    float* zf[] = {iou_scores, class_mask, tx, ty, tw, th, tconf};
    for (float* p : zf) ODPL_HIP_TRY(odpl::fill_bytes(p, 0, cells * sizeof(float), stream));
    ODPL_HIP_TRY(odpl::fill_bytes(tcls, 0, cells * C * sizeof(float), stream));
    BTParams P;
    P.pred_boxes = pred_boxes; P.pred_cls = pred_cls; P.target = target;
    P.B = B; P.A = A; P.G = G; P.C = C; P.nT = nT; P.thr = ignore_thres; P.state = state;
    P.iou_scores = iou_scores; P.class_mask = class_mask; P.obj_mask = obj_mask; P.noobj_mask = noobj_mask;
    P.tx = tx; P.ty = ty; P.tw = tw; P.th = th; P.tcls = tcls; P.tconf = tconf; P.status = status;
    hipLaunchKernelGGL(build_targets_kernel, dim3(1), dim3(kThreads), 0, stream, P, anchors);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
