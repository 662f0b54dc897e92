// Backward-data of the 1x1 convolution that opens a residual branch, with the skip path's gradient added in the epilogue.
//
// A residual unit of the reference's backbones (utils/YoloV4Utils.py:238-262 ResBlock: h = x + conv3x3(conv1x1(x)), each with BN + Mish)
// uses its input twice; autograd therefore sums two gradients for x -- the skip's (d loss / d h itself) and the branch's (the 1x1
// convolution's backward-data) -- with an elementwise `add` launch per unit: three passes over the map, 23 units per YOLOv4 step, 0.48 ms
// (profiles/r04_bench_steady_state.csv).  On channels-last tensors that backward-data IS a row-major GEMM, dx[M, Cin] = dy[M, Cout] .
// w[Cout, Cin] with M = N*H*W, so the sum costs one extra read in its epilogue: dx = dy . w + g_skip, one rounding instead of two.
// The GEMM itself is round 3's fp32-MFMA probe (v_mfma_f32_32x32x2_f32: exact fp32 products, a k-ordered sum; 64 x 64 or 128 x 128
// tiles, K step 32, next K tile prefetched into registers, one float of LDS padding per row, XCD-aware tile order), which on the
// residual units' shapes is 12 % faster than MIOpen's kernel + its zero fill (profiles/r03_gemm1x1.jsonl: 555 vs 628 us per step over
// the 22 units) -- here it also absorbs the add.  Deterministic (no atomics).  torch binding: odpl_torch_ext conv1x1_skip.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT = 256;
constexpr int BK = 32;

// C[M, N] = A[M, K] . B[K, N] (+ ADD[M, N]):  A = dy rows, B = the 1x1 weight [Cout, Cin] (reduction-major), C = dx
template <int BM, int BN>
__global__ __launch_bounds__(kT) void resgrad_gemm_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ ADD,
                                                         float* __restrict__ C, int M, int N, int K) {
    constexpr bool B_KMAJOR = false;
    constexpr int TM = BM / 64, TN = BN / 64;                          // MFMA tiles per wave (waves: 2 x 2)
    constexpr int LDA = BK + 1;
    constexpr int LDB = B_KMAJOR ? BK + 1 : BN + 4;                    // [BN][BK+1] or [BK][BN+4]
    constexpr int A_F4 = BM * BK / 4 / kT;                             // float4 loads per thread and K tile (A)
    constexpr int B_F4 = BN * BK / 4 / kT;
    __shared__ float As[BM * LDA];
    __shared__ float Bs[B_KMAJOR ? BN * LDB : BK * LDB];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * (BM / 2), wn = (wv & 1) * (BN / 2);
    // block -> tile.  Workgroups are dealt round-robin over the 8 XCDs (b and b + 8 share one; speed only, never correctness): the
    // workgroups of one XCD walk the N tiles of an M tile first, so an A panel is pulled into ONE XCD's L2 and reused by all of that
    // row's N tiles there (the weights are small and end up in every L2 anyway).  Tiles past the grid's rectangular part fall back to
    // the plain order.
    const int tiles_m = (M + BM - 1) / BM, tiles_n = N / BN;
    int tm, tn;
    {
        const int b = blockIdx.x, x = b & 7, q = b >> 3;
        const int full = (tiles_m / 8) * 8;                              // M tiles covered by whole groups of 8
        if (b < full * tiles_n) { tm = 8 * (q / tiles_n) + x; tn = q - (q / tiles_n) * tiles_n; }
        else { const int r = b - full * tiles_n; tm = full + r % (tiles_m - full); tn = r / (tiles_m - full); }
    }
    const int bm = tm * BM, bn = tn * BN;
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    float4 ra[A_F4], rb[B_F4];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + kT * i, row = f / (BK / 4), kq = f - row * (BK / 4);
            int gr = bm + row; if (gr > M - 1) gr = M - 1;             // clamped: rows past M are computed and never stored
            ra[i] = *reinterpret_cast<const float4*>(A + (size_t)gr * K + k0 + 4 * kq);
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + kT * i;
            if (B_KMAJOR) {
                const int row = f / (BK / 4), kq = f - row * (BK / 4);                  // row = output column n
                rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(bn + row) * K + k0 + 4 * kq);
            } else {
                const int kr = f / (BN / 4), nq = f - kr * (BN / 4);                    // kr = reduction index
                rb[i] = *reinterpret_cast<const float4*>(B + (size_t)(k0 + kr) * N + bn + 4 * nq);
            }
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < A_F4; ++i) {
            const int f = tid + kT * i, row = f / (BK / 4), kq = f - row * (BK / 4);
            float* d = As + row * LDA + 4 * kq;
            d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
        }
#pragma unroll
        for (int i = 0; i < B_F4; ++i) {
            const int f = tid + kT * i;
            if (B_KMAJOR) {
                const int row = f / (BK / 4), kq = f - row * (BK / 4);
                float* d = Bs + row * LDB + 4 * kq;
                d[0] = rb[i].x; d[1] = rb[i].y; d[2] = rb[i].z; d[3] = rb[i].w;
            } else {
                const int kr = f / (BN / 4), nq = f - kr * (BN / 4);
                *reinterpret_cast<float4*>(Bs + kr * LDB + 4 * nq) = rb[i];
            }
        }
    };
    const int nk = K / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    const int lr = lane & 31, lk = lane >> 5;
    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);                    // in flight during the multiplication below
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[(wm + 32 * i + lr) * LDA + kk + lk];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = B_KMAJOR ? Bs[(wn + 32 * j + lr) * LDB + kk + lk] : Bs[(kk + lk) * LDB + wn + 32 * j + lr];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
        if (kt + 1 < nk) { store_tiles(); __syncthreads(); }
    }
    // ---- epilogue: D[row][col], col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = bm + wm + 32 * i + (r & 3) + 8 * (r >> 2) + 4 * lk, n = bn + wn + 32 * j + lr;
                if (m < M) C[(size_t)m * N + n] = ADD ? acc[i][j][r] + ADD[(size_t)m * N + n] : acc[i][j][r];
            }
}
}  // namespace


// dx[M, Cin] = dy[M, Cout] . w[Cout, Cin] (+ add[M, Cin]) ; Cout % 32 == 0, Cin % 64 == 0; tile: 0 = by shape, 64 or 128
extern "C" int odpl_conv1x1_backward_data_add(const float* dy, const float* w, const float* add, float* dx, int64_t M, int32_t Cin, int32_t Cout,
                                              int32_t tile, void* stream_) {
    if (!dy || !w || !dx || M < 1 || M > 0x7fffffffLL || Cout < 32 || (Cout % 32) || Cin < 64 || (Cin % 64)) return ODPL_EINVAL;
    if ((((uintptr_t)dy | (uintptr_t)w | (uintptr_t)dx | (uintptr_t)add) & 15)) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    const bool big = tile ? tile == 128 : (Cin % 128 == 0 && M >= 128 * 64);
    if (big && Cin % 128) return ODPL_EINVAL;
    if (big) {
        const long long blocks = ((M + 127) / 128) * (Cin / 128);
        hipLaunchKernelGGL((resgrad_gemm_kernel<128, 128>), dim3((unsigned)blocks), dim3(kT), 0, stream, dy, w, add, dx, (int)M, Cin, Cout);
    } else {
        const long long blocks = ((M + 63) / 64) * (Cin / 64);
        hipLaunchKernelGGL((resgrad_gemm_kernel<64, 64>), dim3((unsigned)blocks), dim3(kT), 0, stream, dy, w, add, dx, (int)M, Cin, Cout);
    }
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
