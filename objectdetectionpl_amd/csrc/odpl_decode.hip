// YOLOv4 inference decode for MI355X (gfx950): yolo_forward_dynamic + get_region_boxes
// (LightningFunc/utils/YoloV4Utils.py:18-176) and YoloLayer's masked anchors (model/YOLOV4.py:41-49),
// all scales in one launch, written straight into the concatenated outputs.
//
// HBM-bound transpose: the head is channel-planar ([B, A*(5+C), H, W]) while confs is candidate-major
// ([B, N, C]).  A workgroup owns 64 consecutive cells of one (b, a) plane: it reads each of the 5+C
// channel rows as one coalesced 256-B segment per wave-instruction into an LDS tile [5+C][64], then
// writes the 64 x C products sigmoid(cls)*sigmoid(det) as ONE contiguous 64*C*4-byte run (coalesced),
// and the 64 boxes as float4.  Algorithmic bytes: 4*B*N*(5+C) read + 4*B*N*(4+C) written.
#include "odpl_common.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kCellShift = 6;          // 64 cells per workgroup (256-B channel-row segments): 26.9 us; 128 cells (512-B segments, 44 KB LDS) measured 31.0 us
constexpr int kCells = 1 << kCellShift;
constexpr int kThreads = 256;
constexpr int kLoads = 11;

struct DParams {
    const float* head[ODPL_MAX_SCALES];
    int H[ODPL_MAX_SCALES], W[ODPL_MAX_SCALES], A[ODPL_MAX_SCALES], first[ODPL_MAX_SCALES + 1], blk0[ODPL_MAX_SCALES + 1];
    float aw[ODPL_MAX_SCALES][ODPL_MAX_ANCHORS], ah[ODPL_MAX_SCALES][ODPL_MAX_ANCHORS];
    int nscales, B, C, Ntot;
    unsigned magic_c4, magic_c;       // ceil(2^32 / (C/4)), ceil(2^32 / C): the write phase divides tile-local indices < 2^16
    float sxy, sxy_off;               // scale_x_y and 0.5*(scale_x_y-1), YoloV4Utils.py:78
    float* boxes; float* confs;
};

__global__ __launch_bounds__(kThreads) void yolo_decode_kernel(const DParams P) {
    extern __shared__ float s_t[];                       // [(5+C)][kCells+1]
    const int tid = threadIdx.x, b = blockIdx.y, C = P.C, CH = 5 + C, ld = kCells + 1;
    int s = 0;
    while (s + 1 < P.nscales && (int)blockIdx.x >= P.blk0[s + 1]) ++s;
    const int HW = P.H[s] * P.W[s], tiles = (HW + kCells - 1) / kCells;
    const int lb = blockIdx.x - P.blk0[s], a = lb / tiles, sp0 = (lb - a * tiles) * kCells;
    const int nc = (HW - sp0) < kCells ? (HW - sp0) : kCells;
    const float* src = P.head[s] + ((size_t)(b * P.A[s] + a) * CH) * HW + sp0;
    if ((HW & 3) == 0 && nc == kCells) {
        // full tile of a scale whose channel rows are 16-byte aligned (52^2, 26^2): 16-B loads, 16 lanes per channel row -- a
        // quarter of the memory instructions of the 4-byte walk below (the request rate, not the bytes, bounded the read phase)
        constexpr int kQ = kCells / 4, kVL = 6;                      // float4 per row; loads in flight per lane (C = 80: 1360 / 256 -> one trip)
        for (int e = tid; e < CH * kQ; e += kVL * kThreads) {
            float4 v[kVL];
#pragma unroll
            for (int u = 0; u < kVL; ++u) {
                const int ee = e + u * kThreads, c = ee / kQ, i4 = ee - c * kQ;
                v[u] = c < CH ? *reinterpret_cast<const float4*>(src + (size_t)c * HW + 4 * i4) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int u = 0; u < kVL; ++u) {
                const int ee = e + u * kThreads, c = ee / kQ, i4 = ee - c * kQ;
                if (c < CH) { float* d = s_t + c * ld + 4 * i4; d[0] = v[u].x; d[1] = v[u].y; d[2] = v[u].z; d[3] = v[u].w; }
            }
        }
    } else {
        for (int e = tid; e < CH * kCells; e += kLoads * kThreads) {   // channel rows: 64 consecutive floats per row; kLoads loads in
            float v[kLoads];                                            // flight per lane (C = 80: the whole tile in two trips)
#pragma unroll
            for (int u = 0; u < kLoads; ++u) {
                const int ee = e + u * kThreads, c = ee >> kCellShift, i = ee & (kCells - 1);
                v[u] = (c < CH && i < nc) ? src[(size_t)c * HW + i] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < kLoads; ++u) {
                const int ee = e + u * kThreads, c = ee >> kCellShift, i = ee & (kCells - 1);
                if (c < CH && i < nc) s_t[c * ld + i] = v[u];
            }
        }
    }
    __syncthreads();
    const size_t n0 = (size_t)b * P.Ntot + P.first[s] + (size_t)a * HW + sp0;     // first candidate of the tile
    if (tid < nc) {
        const int sp = sp0 + tid, gy = sp / P.W[s], gx = sp - gy * P.W[s];
        // YoloV4Utils.py:78-79, 110-145
        const float bx = (sigmoid_fast(s_t[0 * ld + tid]) * P.sxy - P.sxy_off + (float)gx) / (float)P.W[s];
        const float by = (sigmoid_fast(s_t[1 * ld + tid]) * P.sxy - P.sxy_off + (float)gy) / (float)P.H[s];
        const float bw = (expf(s_t[2 * ld + tid]) * P.aw[s][a]) / (float)P.W[s];
        const float bh = (expf(s_t[3 * ld + tid]) * P.ah[s][a]) / (float)P.H[s];
        const float x1 = bx - bw * 0.5f, y1 = by - bh * 0.5f;                     // :153-156
        reinterpret_cast<float4*>(P.boxes)[n0 + tid] = make_float4(x1, y1, x1 + bw, y1 + bh);
        s_t[4 * ld + tid] = sigmoid_fast(s_t[4 * ld + tid]);                         // det conf, reused below
    }
    __syncthreads();
    float* dst = P.confs + n0 * C;                       // 64 candidates x C classes, contiguous
    if ((C & 3) == 0 && ((n0 * C) & 3) == 0) {           // 16-B stores: 4 consecutive classes of one candidate
        const int c4 = C >> 2;
        for (int e = tid; e < nc * c4; e += kThreads) {
            const int i = fastdiv16(e, P.magic_c4), k = 4 * (e - i * c4);
            const float dc = s_t[4 * ld + i];
            float4 o;
            o.x = sigmoid_fast(s_t[(5 + k) * ld + i]) * dc; o.y = sigmoid_fast(s_t[(6 + k) * ld + i]) * dc;      // :170
            o.z = sigmoid_fast(s_t[(7 + k) * ld + i]) * dc; o.w = sigmoid_fast(s_t[(8 + k) * ld + i]) * dc;
            reinterpret_cast<float4*>(dst)[e] = o;
        }
    } else {
        for (int e = tid; e < nc * C; e += kThreads) {
            const int i = fastdiv16(e, P.magic_c), k = e - i * C;
            dst[e] = sigmoid_fast(s_t[(5 + k) * ld + i]) * s_t[4 * ld + i];       // :170
        }
    }
}
}  // namespace

extern "C" int odpl_yolo_decode(const odpl_yolo_decode_desc_t* d, float* boxes, float* confs, void* stream_) {
    if (!d || !boxes || !confs || d->nscales < 1 || d->nscales > ODPL_MAX_SCALES || d->B < 1 || d->C < 1) return ODPL_EINVAL;
    if ((uintptr_t)boxes & 15) return ODPL_EINVAL;
    if ((size_t)(5 + d->C) * (kCells + 1) * 4 > 150 * 1024 || d->C > 1023) return ODPL_EINVAL;      // kCells * C < 2^16 (fastdiv16)
    DParams P;
    int first = 0, blk = 0;
    for (int i = 0; i < ODPL_MAX_SCALES; ++i) {
        P.first[i] = first; P.blk0[i] = blk;
        if (i < d->nscales) {
            const odpl_yolo_decode_scale_t& s = d->scale[i];
            if (!s.head || s.H < 1 || s.W < 1 || s.A < 1 || s.A > ODPL_MAX_ANCHORS || ((uintptr_t)s.head & 15)) return ODPL_EINVAL;   // 16-B loads on aligned scales
            P.head[i] = s.head; P.H[i] = s.H; P.W[i] = s.W; P.A[i] = s.A;
            for (int a = 0; a < ODPL_MAX_ANCHORS; ++a) { P.aw[i][a] = s.anchor_w[a]; P.ah[i][a] = s.anchor_h[a]; }
            first += s.A * s.H * s.W;
            blk += s.A * ((s.H * s.W + kCells - 1) / kCells);
        } else { P.head[i] = nullptr; P.H[i] = P.W[i] = P.A[i] = 1; }
    }
    P.first[ODPL_MAX_SCALES] = first; P.blk0[ODPL_MAX_SCALES] = blk;
    P.nscales = d->nscales; P.B = d->B; P.C = d->C; P.Ntot = first;
    P.sxy = d->scale_x_y; P.sxy_off = d->scale_x_y_offset;
    P.boxes = boxes; P.confs = confs;
    P.magic_c = fastdiv16_magic((unsigned)d->C); P.magic_c4 = fastdiv16_magic((unsigned)(d->C >> 2));
    hipStream_t stream = (hipStream_t)stream_;
    prof_begin(ODPL_PROF_YOLO_DECODE, stream);
    hipLaunchKernelGGL(yolo_decode_kernel, dim3(blk, d->B), dim3(kThreads), (size_t)(5 + d->C) * (kCells + 1) * 4, stream, P);
    prof_end(ODPL_PROF_YOLO_DECODE, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
