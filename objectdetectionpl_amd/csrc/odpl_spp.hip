// SPP block of the YOLOv4 neck (LightningFunc/utils/YoloV4Utils.py:400-420: three stride-1 max-pools 5/9/13 and
// the concat [m13, m9, m5, x]) as one forward and one backward kernel -- SURVEY 8f rank 4 ("SPP max-pools").
// ATen's max_pool2d backward on channels-last tensors costs ~1 ms per call here (3 calls/step = 6 % of the YOLOv4
// step); this pair takes a few tens of microseconds.
//   forward : one workgroup per (b, c) plane: plane -> LDS, each thread takes the 5x5 / 9x9 / 13x13 window maxima
//             of its position (nested windows share the scan) and writes the four channel blocks of the output.
//   backward: plane -> LDS, arg-max position of every window recomputed into LDS, then a GATHER: every input
//             position sums the gradients of the windows that selected it -- no atomics, deterministic.
// Arbitrary (dense) strides: works for NCHW and channels-last tensors alike.  First maximum in row-major window
// order wins ties and a NaN wins, like ATen.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kMaxHW = 1024;          // plane positions held in LDS (32 x 32)

struct SppParams {
    const float* x; float* y; const float* gy; float* gx;
    int B, C, H, W;
    long long xs[4], ys[4];           // element strides (b, c, h, w) of x / gx and of y / gy
};

__device__ __forceinline__ void window_scan(const float* s, int H, int W, int py, int px, float* mx, int* am) {
    // nested windows r = 2, 4, 6: visit the 13x13 window once in row-major order and update each level whose
    // window contains the element -- first maximum wins, NaN wins and sticks
    float m0 = 0.f, m1 = 0.f, m2 = 0.f; int a0 = -1, a1 = -1, a2 = -1;
    for (int dy = -6; dy <= 6; ++dy) {
        const int yy = py + dy;
        if (yy < 0 || yy >= H) continue;
        for (int dx = -6; dx <= 6; ++dx) {
            const int xx = px + dx;
            if (xx < 0 || xx >= W) continue;
            const int q = yy * W + xx;
            const float v = s[q];
            const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
            if (a2 < 0 || (m2 == m2 && (v > m2 || v != v))) { m2 = v; a2 = q; }
            if (ady <= 4 && adx <= 4 && (a1 < 0 || (m1 == m1 && (v > m1 || v != v)))) { m1 = v; a1 = q; }
            if (ady <= 2 && adx <= 2 && (a0 < 0 || (m0 == m0 && (v > m0 || v != v)))) { m0 = v; a0 = q; }
        }
    }
    mx[0] = m0; mx[1] = m1; mx[2] = m2; am[0] = a0; am[1] = a1; am[2] = a2;
}

__global__ __launch_bounds__(256) void spp_fwd_kernel(const SppParams P) {
    __shared__ float s[kMaxHW];
    const int c = blockIdx.x, b = blockIdx.y, HW = P.H * P.W;
    const float* xp = P.x + b * P.xs[0] + c * P.xs[1];
    for (int q = threadIdx.x; q < HW; q += 256) s[q] = xp[(q / P.W) * P.xs[2] + (q % P.W) * P.xs[3]];
    __syncthreads();
    for (int q = threadIdx.x; q < HW; q += 256) {
        float mx[3]; int am[3];
        const int py = q / P.W, px = q - py * P.W;
        window_scan(s, P.H, P.W, py, px, mx, am);
        float* yp = P.y + b * P.ys[0] + py * P.ys[2] + px * P.ys[3];
        yp[(long long)c * P.ys[1]] = mx[2];                          // [m13 | m9 | m5 | x]  (YoloV4Utils.py:418)
        yp[(long long)(P.C + c) * P.ys[1]] = mx[1];
        yp[(long long)(2 * P.C + c) * P.ys[1]] = mx[0];
        yp[(long long)(3 * P.C + c) * P.ys[1]] = s[q];
    }
}

__global__ __launch_bounds__(256) void spp_bwd_kernel(const SppParams P) {
    __shared__ float s[kMaxHW];
    __shared__ float g[3][kMaxHW];
    __shared__ short am[3][kMaxHW];
    const int c = blockIdx.x, b = blockIdx.y, HW = P.H * P.W;
    const float* xp = P.x + b * P.xs[0] + c * P.xs[1];
    for (int q = threadIdx.x; q < HW; q += 256) s[q] = xp[(q / P.W) * P.xs[2] + (q % P.W) * P.xs[3]];
    __syncthreads();
    for (int q = threadIdx.x; q < HW; q += 256) {
        float mx[3]; int a[3];
        const int py = q / P.W, px = q - py * P.W;
        window_scan(s, P.H, P.W, py, px, mx, a);
        const float* gp = P.gy + b * P.ys[0] + py * P.ys[2] + px * P.ys[3];
        am[0][q] = (short)a[0]; am[1][q] = (short)a[1]; am[2][q] = (short)a[2];
        g[2][q] = gp[(long long)c * P.ys[1]];                        // gradient of the 13-pool block
        g[1][q] = gp[(long long)(P.C + c) * P.ys[1]];
        g[0][q] = gp[(long long)(2 * P.C + c) * P.ys[1]];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int py = p / P.W, px = p - py * P.W;
        float acc = P.gy[b * P.ys[0] + (long long)(3 * P.C + c) * P.ys[1] + py * P.ys[2] + px * P.ys[3]];   // identity branch
        for (int dy = -6; dy <= 6; ++dy) {                           // windows that contain p, fixed order
            const int yy = py + dy;
            if (yy < 0 || yy >= P.H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = px + dx;
                if (xx < 0 || xx >= P.W) continue;
                const int q = yy * P.W + xx;
                const int ady = dy < 0 ? -dy : dy, adx = dx < 0 ? -dx : dx;
                if (am[2][q] == p) acc += g[2][q];
                if (ady <= 4 && adx <= 4 && am[1][q] == p) acc += g[1][q];
                if (ady <= 2 && adx <= 2 && am[0][q] == p) acc += g[0][q];
            }
        }
        P.gx[b * P.xs[0] + c * P.xs[1] + py * P.xs[2] + px * P.xs[3]] = acc;
    }
}

bool fill(SppParams& P, const float* x, const int64_t* xs, int B, int C, int H, int W, const int64_t* ys) {
    if (!x || !xs || !ys || B < 1 || C < 1 || H < 1 || W < 1 || H * W > kMaxHW || B > 65535) return false;
    P.x = x; P.B = B; P.C = C; P.H = H; P.W = W;
    for (int i = 0; i < 4; ++i) { P.xs[i] = xs[i]; P.ys[i] = ys[i]; }
    return true;
}
}  // namespace

extern "C" int odpl_spp_forward(const float* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                float* y, const int64_t* y_strides, void* stream) {
    SppParams P;
    if (!y || !fill(P, x, x_strides, B, C, H, W, y_strides)) return ODPL_EINVAL;
    P.y = y; P.gy = nullptr; P.gx = nullptr;
    hipLaunchKernelGGL(spp_fwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_spp_backward(const float* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                 const float* grad_y, const int64_t* y_strides, float* grad_x, void* stream) {
    SppParams P;
    if (!grad_y || !grad_x || !fill(P, x, x_strides, B, C, H, W, y_strides)) return ODPL_EINVAL;
    P.y = nullptr; P.gy = grad_y; P.gx = grad_x;
    hipLaunchKernelGGL(spp_bwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
