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

// Focus on an RGB image with channels-last memory on both sides (the YOLOv5 stem as the caller networks run it): an output
// pixel is 12 contiguous floats gathered from two runs of 6 contiguous floats (pixels 2j, 2j+1 of rows 2i and 2i+1), so one
// thread moves one output pixel with three 8-byte loads per row and three 16-byte stores -- 96 B per thread, every access
// contiguous across lanes.  (The strided kernel above spends three 64-bit divisions per 8 bytes and scatters 4-byte stores
// 48 B apart: 264 us for the 157 + 157 MB of BASELINE config 5.)  INV walks the same pairs backwards.
template <bool INV>
__global__ __launch_bounds__(kT) void focus_nhwc3_kernel(const S2dParams P) {
    const int H2 = P.H >> 1, W2 = P.W >> 1;
    const int row = blockIdx.y;                                          // b * H2 + i
    const int b = row / H2, i = row - b * H2;
    const float* xsrc = INV ? nullptr : P.src; float* xdst = INV ? P.dst : nullptr;
    const size_t x0 = ((size_t)(b * P.H + 2 * i) * P.W) * 3, x1 = x0 + (size_t)P.W * 3, y0 = ((size_t)row * W2) * 12;
    for (int j = blockIdx.x * kT + threadIdx.x; j < W2; j += gridDim.x * kT) {
        if (!INV) {
            const float2* r0 = reinterpret_cast<const float2*>(xsrc + x0 + 6 * (size_t)j);
            const float2* r1 = reinterpret_cast<const float2*>(xsrc + x1 + 6 * (size_t)j);
            const float2 a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
            float4* o = reinterpret_cast<float4*>(P.dst + y0 + 12 * (size_t)j);
            o[0] = make_float4(a0.x, a0.y, a1.x, b0.x);                  // g = 0: (2i, 2j)      | g = 1: (2i+1, 2j) ...
            o[1] = make_float4(b0.y, b1.x, a1.y, a2.x);                  // ... g = 1            | g = 2: (2i, 2j+1) ...
            o[2] = make_float4(a2.y, b1.y, b2.x, b2.y);                  // ... g = 2            | g = 3: (2i+1, 2j+1)
        } else {
            const float4* o = reinterpret_cast<const float4*>(P.src + y0 + 12 * (size_t)j);
            const float4 u = o[0], v = o[1], w = o[2];
            float2* r0 = reinterpret_cast<float2*>(xdst + x0 + 6 * (size_t)j);
            float2* r1 = reinterpret_cast<float2*>(xdst + x1 + 6 * (size_t)j);
            r0[0] = make_float2(u.x, u.y); r0[1] = make_float2(u.z, v.z); r0[2] = make_float2(v.w, w.x);
            r1[0] = make_float2(u.w, v.x); r1[1] = make_float2(v.y, w.y); r1[2] = make_float2(w.z, w.w);
        }
    }
}

// Forward, even W/2: one thread moves TWO output pixels -- 48 contiguous bytes per input row as three 16-byte loads, 96 contiguous
// output bytes as six 16-byte stores.
__global__ __launch_bounds__(kT) void focus_nhwc3x2_kernel(const S2dParams P) {
    const int H2 = P.H >> 1, W4 = P.W >> 2;
    const int row = blockIdx.y;
    const int b = row / H2, i = row - b * H2;
    const size_t x0 = ((size_t)(b * P.H + 2 * i) * P.W) * 3, x1 = x0 + (size_t)P.W * 3, y0 = ((size_t)row * (P.W >> 1)) * 12;
    for (int j = blockIdx.x * kT + threadIdx.x; j < W4; j += gridDim.x * kT) {
        const float4* r0 = reinterpret_cast<const float4*>(P.src + x0 + 12 * (size_t)j);
        const float4* r1 = reinterpret_cast<const float4*>(P.src + x1 + 12 * (size_t)j);
        const float4 a0 = r0[0], a1 = r0[1], a2 = r0[2], b0 = r1[0], b1 = r1[1], b2 = r1[2];
        float4* o = reinterpret_cast<float4*>(P.dst + y0 + 24 * (size_t)j);
        // row floats 0..11 = pixels 4j .. 4j+3 (3 channels each): a0 = (0,1,2,3) a1 = (4,5,6,7) a2 = (8,9,10,11)
        o[0] = make_float4(a0.x, a0.y, a0.z, b0.x);          // pixel 2j:   [p0 row0 | p0 row1 | p1 row0 | p1 row1]
        o[1] = make_float4(b0.y, b0.z, a0.w, a1.x);
        o[2] = make_float4(a1.y, b0.w, b1.x, b1.y);
        o[3] = make_float4(a1.z, a1.w, a2.x, b1.z);          // pixel 2j+1: [p2 row0 | p2 row1 | p3 row0 | p3 row1]
        o[4] = make_float4(b1.w, b2.x, a2.y, a2.z);
        o[5] = make_float4(a2.w, b2.y, b2.z, b2.w);
    }
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
    const long long H2 = H / 2, W2 = W / 2;
    const bool x_cl = C == 3 && P.xs[1] == 1 && P.xs[3] == 3 && P.xs[2] == 3LL * W && P.xs[0] == 3LL * W * H;
    const bool y_cl = P.ys[1] == 1 && P.ys[3] == 12 && P.ys[2] == 12 * W2 && P.ys[0] == 12 * W2 * H2;
    const void* xp = inverse ? (const void*)dst : (const void*)src; const void* yp = inverse ? (const void*)src : (const void*)dst;
    if (x_cl && y_cl && (long long)B * H2 <= 65535 && !((uintptr_t)xp & 7) && !((uintptr_t)yp & 15)) {
        if (!inverse && (W2 & 1) == 0 && !((uintptr_t)xp & 15)) {
            int bx2 = (int)((W2 / 2 + kT - 1) / kT); if (bx2 > 64) bx2 = 64;
            hipLaunchKernelGGL(focus_nhwc3x2_kernel, dim3(bx2, (unsigned)(B * H2)), dim3(kT), 0, (hipStream_t)stream_, P);
            ODPL_HIP_TRY(hipGetLastError());
            return 0;
        }
        int bx = (int)((W2 + kT - 1) / kT); if (bx > 64) bx = 64;
        const dim3 grid(bx, (unsigned)(B * H2));
        if (inverse) hipLaunchKernelGGL(focus_nhwc3_kernel<true>, grid, dim3(kT), 0, (hipStream_t)stream_, P);
        else hipLaunchKernelGGL(focus_nhwc3_kernel<false>, grid, dim3(kT), 0, (hipStream_t)stream_, P);
        ODPL_HIP_TRY(hipGetLastError());
        return 0;
    }
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
