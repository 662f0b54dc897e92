// get_yolo_statistics (accuracy.py:382-470) for one scale: training-style decode of the raw head into
// pred_boxes / pred_conf / pred_cls (+ the concatenated `output` tensor), then -- after the dense build_targets
// entry (odpl_yolo_build_targets) has produced the masks -- the six scalar metrics.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kCells = 64, kThreads = 256;

struct SDParams {
    const float* head; int B, A, C, G; float aw[ODPL_MAX_ANCHORS], ah[ODPL_MAX_ANCHORS]; float stride;
    float* pred_boxes; float* pred_conf; float* pred_cls; float* output;
};

__global__ __launch_bounds__(kThreads) void yolo_train_decode_kernel(const SDParams P) {
    extern __shared__ float s_t[];                       // [(5+C)][kCells+1]
    const int tid = threadIdx.x, b = blockIdx.y, C = P.C, CH = 5 + C, ld = kCells + 1;
    const int HW = P.G * P.G, tiles = (HW + kCells - 1) / kCells;
    const int a = blockIdx.x / tiles, sp0 = (blockIdx.x - a * tiles) * kCells;
    const int nc = (HW - sp0) < kCells ? (HW - sp0) : kCells;
    const float* src = P.head + ((size_t)(b * P.A + a) * CH) * HW + sp0;
    for (int e = tid; e < CH * kCells; e += kThreads) {
        const int c = e >> 6, i = e & 63;
        if (i < nc) s_t[c * ld + i] = src[(size_t)c * HW + i];
    }
    __syncthreads();
    const size_t cell0 = (size_t)(b * P.A + a) * HW + sp0;
    if (tid < nc) {
        const int sp = sp0 + tid, gy = sp / P.G, gx = sp - gy * P.G;
        const float px = sigmoidf_(s_t[0 * ld + tid]) + (float)gx;           // accuracy.py:412-435
        const float py = sigmoidf_(s_t[1 * ld + tid]) + (float)gy;
        const float pw = expf(s_t[2 * ld + tid]) * P.aw[a], ph = expf(s_t[3 * ld + tid]) * P.ah[a];
        const float cf = sigmoidf_(s_t[4 * ld + tid]);
        reinterpret_cast<float4*>(P.pred_boxes)[cell0 + tid] = make_float4(px, py, pw, ph);
        P.pred_conf[cell0 + tid] = cf;
        float* o = P.output + (cell0 + tid) * CH;                            // accuracy.py:459-466
        o[0] = px * P.stride; o[1] = py * P.stride; o[2] = pw * P.stride; o[3] = ph * P.stride; o[4] = cf;
    }
    for (int e = tid; e < nc * C; e += kThreads) {
        const int i = e / C, k = e - i * C;
        const float v = sigmoidf_(s_t[(5 + k) * ld + i]);
        P.pred_cls[(cell0 + i) * C + k] = v;
        P.output[(cell0 + i) * CH + 5 + k] = v;
    }
}

__global__ __launch_bounds__(1024) void yolo_stats_reduce_kernel(const float* __restrict__ conf, const float* __restrict__ class_mask,
                                                                 const float* __restrict__ iou, const unsigned char* __restrict__ obj,
                                                                 const unsigned char* __restrict__ noobj, int cells, float* __restrict__ out6) {
    __shared__ float sh[8 * 16];
    const int tid = threadIdx.x;
    float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // sum cm[obj], nObj, sum conf[obj], sum conf[noobj], nNoobj, sum conf50, sum iou50*det, sum iou75*det
    for (int i = tid; i < cells; i += 1024) {
        const float c = conf[i], cm = class_mask[i], io = iou[i];
        const float o = obj[i] ? 1.f : 0.f, n = noobj[i] ? 1.f : 0.f;
        const float c50 = c > 0.5f ? 1.f : 0.f;
        const float det = c50 * cm * o;                                      // accuracy.py:454 (tconf == obj_mask)
        a[0] += cm * o; a[1] += o; a[2] += c * o; a[3] += c * n; a[4] += n; a[5] += c50;
        a[6] += (io > 0.5f ? 1.f : 0.f) * det; a[7] += (io > 0.75f ? 1.f : 0.f) * det;
    }
    block_sum<8>(a, sh);
    if (tid == 0) {
        out6[0] = 100.0f * (a[0] / a[1]);               // cls_acc  (mean over obj cells; NaN if none, as the reference)
        out6[1] = a[6] / (a[1] + 1e-16f);               // recall50
        out6[2] = a[7] / (a[1] + 1e-16f);               // recall75
        out6[3] = a[6] / (a[5] + 1e-16f);               // precision
        out6[4] = a[2] / a[1];                          // conf_obj
        out6[5] = a[3] / a[4];                          // conf_noobj
    }
}
}  // namespace

extern "C" int odpl_yolo_train_decode(const float* head, int32_t B, int32_t A, int32_t C, int32_t G, const float* host_anchor_wh,
                                      float stride, float* pred_boxes, float* pred_conf, float* pred_cls, float* output, void* stream) {
    if (!head || !host_anchor_wh || !pred_boxes || !pred_conf || !pred_cls || !output || B < 1 || A < 1 || A > ODPL_MAX_ANCHORS || C < 1 || G < 1)
        return ODPL_EINVAL;
    if (((uintptr_t)pred_boxes & 15) || (size_t)(5 + C) * (kCells + 1) * 4 > 60 * 1024) return ODPL_EINVAL;
    SDParams P;
    P.head = head; P.B = B; P.A = A; P.C = C; P.G = G; P.stride = stride;
    for (int a = 0; a < A; ++a) { P.aw[a] = host_anchor_wh[2 * a]; P.ah[a] = host_anchor_wh[2 * a + 1]; }
    P.pred_boxes = pred_boxes; P.pred_conf = pred_conf; P.pred_cls = pred_cls; P.output = output;
    const int tiles = (G * G + kCells - 1) / kCells;
    hipLaunchKernelGGL(yolo_train_decode_kernel, dim3(A * tiles, B), dim3(kThreads), (size_t)(5 + C) * (kCells + 1) * 4, (hipStream_t)stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_yolo_stats_reduce(const float* pred_conf, const float* class_mask, const float* iou_scores, const uint8_t* obj_mask,
                                      const uint8_t* noobj_mask, int64_t cells, float* out6, void* stream) {
    if (!pred_conf || !class_mask || !iou_scores || !obj_mask || !noobj_mask || !out6 || cells < 1 || cells > 0x7fffffffLL) return ODPL_EINVAL;
    hipLaunchKernelGGL(yolo_stats_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, pred_conf, class_mask, iou_scores, obj_mask, noobj_mask, (int)cells, out6);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
