// The two space-to-depth shuffles on the caller side of the heads, pure HBM data movement:
//   Focus (YOLOv5 stem, utils/YoloV5Utils.py:117-125):
//       y[b, g*C + c, i, j] = x[b, c, 2i + (g & 1), 2j + (g >> 1)],  g = 0..3   (cat of the four ::2 / 1::2 slices)
//   reorg (YOLOv2 passthrough, model/YOLOV2.py:147-151): the reference REINTERPRETS the contiguous [B, C, H, W] block as
//       [B, C/4, H, 2, W, 2], permutes to [B, 2, 2, C/4, H, W] and views that as [B, 4C, H/2, W/2]; per image
//       y_flat[((p*2 + q)*C/4 + c')*H*W + h*W + w] = x_flat[(((c'*H + h)*2 + p)*W + w)*2 + q]
//       (not a spatial 2x2 split -- reproduced as is).
// Both are permutations, so backward is the same walk with source and destination swapped.  One thread moves one
// 2-wide source pair (both column phases), so every source line is read once; ALGORITHMIC bytes = 2 x 4 B x elements.
// Element strides make NCHW and channels-last tensors (either side) work alike.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;

struct S2dParams {
    const float* src; float* dst;
    int B, C, H, W;                    // shape of the LARGE-spatial tensor x [B, C, H, W]
    long long xs[4], ys[4];            // element strides of x-side and y-side tensors
};

// INV = false: x -> y (forward);  INV = true: grad_y -> grad_x (backward)
template <bool INV>
__global__ __launch_bounds__(kT) void focus_kernel(const S2dParams P) {
    const int H2 = P.H >> 1, W2 = P.W >> 1;
    const long long n = (long long)P.B * P.C * P.H * W2;                  // one thread per (b, c, row, column pair)
    const long long stride = (long long)gridDim.x * kT;
    for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < n; t += stride) {
        const int j = (int)(t % W2); long long r = t / W2;
        const int row = (int)(r % P.H); r /= P.H;
        const int c = (int)(r % P.C); const int b = (int)(r / P.C);
        const int i = row >> 1, dy = row & 1;
        const long long xo = b * P.xs[0] + c * P.xs[1] + row * P.xs[2] + (2LL * j) * P.xs[3];
        const long long y0 = b * P.ys[0] + (long long)(dy * P.C + c) * P.ys[1] + i * P.ys[2] + j * P.ys[3];   // g = dy     (dx = 0)
        const long long y1 = y0 + 2LL * P.C * P.ys[1];                                                        // g = dy + 2 (dx = 1)
        if (!INV) {
            float v0, v1;
            if (P.xs[3] == 1) { const float2 v = *reinterpret_cast<const float2*>(P.src + xo); v0 = v.x; v1 = v.y; }
            else { v0 = P.src[xo]; v1 = P.src[xo + P.xs[3]]; }
            P.dst[y0] = v0; P.dst[y1] = v1;
        } else {
            const float v0 = P.src[y0], v1 = P.src[y1];
            if (P.xs[3] == 1) *reinterpret_cast<float2*>(P.dst + xo) = make_float2(v0, v1);
            else { P.dst[xo] = v0; P.dst[xo + P.xs[3]] = v1; }
        }
    }
    (void)H2;
}

template <bool INV>
__global__ __launch_bounds__(kT) void reorg_kernel(const S2dParams P) {
    // x side is the contiguous NCHW block (the reference's .contiguous().view()), y side strided [B, 4C, H/2, W/2]
    const int Cq = P.C >> 2, H2 = P.H >> 1, W2 = P.W >> 1;
    const long long per_img = (long long)P.C * P.H * P.W, n = (long long)P.B * (per_img >> 1);
    const long long stride = (long long)gridDim.x * kT;
    const long long hw = (long long)P.H * P.W, hw2 = (long long)H2 * W2;
    for (long long t = (long long)blockIdx.x * kT + threadIdx.x; t < n; t += stride) {
        const long long b = t / (per_img >> 1), f2 = t - b * (per_img >> 1);      // f2 = flat source index / 2 (q = 0, 1)
        const int w = (int)(f2 % P.W); long long r = f2 / P.W;
        const int p = (int)(r & 1); r >>= 1;
        const int h = (int)(r % P.H); const int cq = (int)(r / P.H);
        const long long xo = b * per_img + 2 * f2;
        float v[2];
        if (!INV) { const float2 s = *reinterpret_cast<const float2*>(P.src + xo); v[0] = s.x; v[1] = s.y; }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const long long yf = ((long long)(p * 2 + q) * Cq + cq) * hw + (long long)h * P.W + w;   // flat index in [4C, H/2, W/2]
            const long long ch = yf / hw2, rem = yf - ch * hw2;
            const long long yo = b * P.ys[0] + ch * P.ys[1] + (rem / W2) * P.ys[2] + (rem % W2) * P.ys[3];
            if (!INV) P.dst[yo] = v[q]; else v[q] = P.src[yo];
        }
        if (INV) *reinterpret_cast<float2*>(P.dst + xo) = make_float2(v[0], v[1]);
    }
}

bool fill(S2dParams& P, int B, int C, int H, int W, const int64_t* xs, const int64_t* ys) {
    if (B < 1 || C < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || !xs || !ys) return false;
    P.B = B; P.C = C; P.H = H; P.W = W;
    for (int i = 0; i < 4; ++i) { P.xs[i] = xs[i]; P.ys[i] = ys[i]; if (xs[i] < 0 || ys[i] < 0) return false; }
    return true;
}
unsigned grid_for(long long n) { long long b = (n + kT - 1) / kT; if (b > 16384) b = 16384; if (b < 1) b = 1; return (unsigned)b; }
}  // namespace

extern "C" int odpl_focus(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                          const int64_t* host_x_strides, const int64_t* host_y_strides, int32_t inverse, void* stream_) {
    S2dParams P = {};
    if (!src || !dst || !fill(P, B, C, H, W, host_x_strides, host_y_strides)) return ODPL_EINVAL;
    if (P.xs[3] == 1 && ((P.xs[0] | P.xs[1] | P.xs[2]) & 1)) return ODPL_EINVAL;            // float2 pairs must be 8-byte aligned
    if (P.xs[3] == 1 && ((uintptr_t)(inverse ? (const void*)dst : (const void*)src) & 7)) return ODPL_EINVAL;
    P.src = src; P.dst = dst;
    const long long n = (long long)B * C * H * (W / 2);
    if (inverse) hipLaunchKernelGGL(focus_kernel<true>, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream_, P);
    else hipLaunchKernelGGL(focus_kernel<false>, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream_, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_reorg(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                          const int64_t* host_y_strides, int32_t inverse, void* stream_) {
    S2dParams P = {};
    const int64_t xs[4] = {(int64_t)C * H * W, (int64_t)H * W, W, 1};
    if (!src || !dst || (C & 3) || !fill(P, B, C, H, W, xs, host_y_strides)) return ODPL_EINVAL;
    if ((uintptr_t)(inverse ? (const void*)dst : (const void*)src) & 7) return ODPL_EINVAL;
    P.src = src; P.dst = dst;
    const long long n = (long long)B * C * H * W / 2;
    if (inverse) hipLaunchKernelGGL(reorg_kernel<true>, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream_, P);
    else hipLaunchKernelGGL(reorg_kernel<false>, dim3(grid_for(n)), dim3(kT), 0, (hipStream_t)stream_, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
