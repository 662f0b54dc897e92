// SPP block of the YOLOv4 neck (LightningFunc/utils/YoloV4Utils.py:400-420: three stride-1 max-pools 5/9/13 and
// the concat [m13, m9, m5, x]) as one forward and one backward kernel -- SURVEY 8f rank 4 ("SPP max-pools").
// ATen's max_pool2d backward on channels-last tensors costs ~1 ms per call here (3 calls/step = 6 % of the YOLOv4
// step); this pair takes a few tens of microseconds.
//   forward : one workgroup per (b, c) plane: plane -> LDS, each thread takes the 5x5 / 9x9 / 13x13 window maxima
//             of its position (nested windows share the scan) and writes the four channel blocks of the output.
//   backward: plane -> LDS, arg-max position of every window recomputed into LDS, then a GATHER: every input
//             position sums the gradients of the windows that selected it -- no atomics, deterministic.
// Arbitrary (dense) strides: works for NCHW and channels-last tensors alike.  First maximum in row-major window
// order wins ties; a NaN wins (the last one in the window), like ATen.
#include <type_traits>
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kMaxHW = 1024;          // plane positions held in LDS (32 x 32)

struct SppParams {
    const float* x; float* y; const float* gy; float* gx;   // tile kernels: may be bf16 (template BF), addressed through void*
    int blk[4];                       // channel block (0..3) of x, m5, m9, m13 in y: v4 [m13|m9|m5|x] = {3,2,1,0}, v5 [x|m5|m9|m13] = {0,1,2,3}
    int B, C, H, W;
    unsigned magic_w;                 // fastdiv16 magic of W (positions < 1024)
    long long xs[4], ys[4];           // element strides (b, c, h, w) of x / gx and of y / gy
};

__device__ __forceinline__ void window_scan(const float* s, int H, int W, int py, int px, float* mx, int* am) {
    // nested windows r = 2, 4, 6: visit the 13x13 window once in row-major order and update each level whose
    // window contains the element -- first maximum wins; every NaN takes over (ATen's scan)
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
            // ATen: `if ((val > maxval) || isnan(val))` -- strict > keeps the first maximum, EVERY NaN takes over (the last one wins)
            if (a2 < 0 || v != v || v > m2) { m2 = v; a2 = q; }
            if (ady <= 4 && adx <= 4 && (a1 < 0 || v != v || v > m1)) { m1 = v; a1 = q; }
            if (ady <= 2 && adx <= 2 && (a0 < 0 || v != v || v > m0)) { m0 = v; a0 = q; }
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
        yp[(long long)(P.blk[3] * P.C + c) * P.ys[1]] = mx[2];        // v4: [m13 | m9 | m5 | x] (YoloV4Utils.py:418); v5: [x | m5 | m9 | m13]
        yp[(long long)(P.blk[2] * P.C + c) * P.ys[1]] = mx[1];
        yp[(long long)(P.blk[1] * P.C + c) * P.ys[1]] = mx[0];
        yp[(long long)(P.blk[0] * P.C + c) * P.ys[1]] = s[q];
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
        g[2][q] = gp[(long long)(P.blk[3] * P.C + c) * P.ys[1]];      // gradient of the 13-pool block
        g[1][q] = gp[(long long)(P.blk[2] * P.C + c) * P.ys[1]];
        g[0][q] = gp[(long long)(P.blk[1] * P.C + c) * P.ys[1]];
    }
    __syncthreads();
    for (int p = threadIdx.x; p < HW; p += 256) {
        const int py = p / P.W, px = p - py * P.W;
        float acc = P.gy[b * P.ys[0] + (long long)(P.blk[0] * P.C + c) * P.ys[1] + py * P.ys[2] + px * P.ys[3]];   // identity branch
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

// ---- channels-last fast path ------------------------------------------------------------------------------------
// The per-plane kernels above read an NHWC tensor with a stride of C floats (one 64-B sector per element).  Here a workgroup
// owns a [H*W positions] x [CT channels] tile: loads and stores are 16-B, 4*CT-byte contiguous per position, lanes run along
// the channels in LDS (bank-conflict free), and the three pools are a CASCADE of separable 5x5 pools
// (maxpool9 = maxpool5 o maxpool5, maxpool13 = maxpool5 o maxpool9 for stride 1 / same padding): 30 LDS reads per output
// instead of 169.  Backward recomputes the cascade on 64-bit winner keys ("NaN (later position) > larger value > smaller
// position" as one unsigned number, see winner_key), records the winning tap of each 1-D pass and routes the gradient back
// along them -- exactly ATen's `if (val > maxval || isnan(val))` row-major arg-max, deterministic, no atomics.
constexpr int kTileT = 1024;           // 16 waves per workgroup: the tile fills the LDS (one workgroup per CU), waves hide its latency

// ---- lean 1-D five-tap passes -------------------------------------------------------------------------------------
// A thread owns the elements (position p_k = pl + k * PL, channel ch), k < KMAX, for the whole kernel: the LDS index of each
// and the validity of its +-2 neighbours along x and y are computed ONCE (the first version re-derived them, with a division,
// in every pass and spent ~150 instructions per element and pass on it and on branchy winner logic).  Taps are loaded
// unconditionally from a clamped address and neutralised by a select, so five independent LDS reads are in flight.
template <int KMAX>
struct Lanes {
    int base[KMAX];           // p * CT + ch, or -1 beyond the tile
    unsigned mask[KMAX];      // bit t: x-tap t (px + t - 2) inside the row; bit 8 + t: y-tap t inside the column
};
template <int CT, int KMAX>
__device__ __forceinline__ void lanes_init(Lanes<KMAX>& L, int H, int W, unsigned mw) {
    const int ch = threadIdx.x % CT, pl = threadIdx.x / CT, HW = H * W;
    constexpr int PL = kTileT / CT;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int p = pl + k * PL;
        L.base[k] = -1; L.mask[k] = 0u;
        if (p < HW) {
            const int py = fastdiv16(p, mw), px = p - py * W;
            unsigned m = 0u;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                m |= (unsigned)(px + t - 2 >= 0 && px + t - 2 < W) << t;
                m |= (unsigned)(py + t - 2 >= 0 && py + t - 2 < H) << (8 + t);
            }
            L.base[k] = p * CT + ch; L.mask[k] = m;
        }
    }
}

// forward values: max over the valid taps, any NaN wins (ATen: `val > maxval || isnan(val)`); ties are irrelevant for the VALUE
template <int KMAX, bool ALONG_X>
__device__ __forceinline__ void max5_pass(const float* src, float* dst, const Lanes<KMAX>& L, int stride) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int b = L.base[k];
        if (b >= 0) {
            const unsigned m = L.mask[k] >> (ALONG_X ? 0 : 8);
            float v[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const bool ok = (m >> t) & 1u;
                const float x = src[ok ? b + (t - 2) * stride : b];
                v[t] = ok ? x : -INFINITY;
            }
            const float r = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[3], v[4])), v[2]);         // fmaxf skips NaNs ...
            const bool nan = (v[0] != v[0]) | (v[1] != v[1]) | (v[2] != v[2]) | (v[3] != v[3]) | (v[4] != v[4]);
            dst[b] = nan ? __uint_as_float(0x7fc00000u) : r;                                   // ... which win
        }
    }
    lds_barrier();
}

// Winner keys of the backward recomputation: (ordered value, tie-break) as ONE 64-bit number whose maximum is ATen's winner --
// NaN above every number, then the larger value, then the SMALLER original position; among NaNs the LARGER position (the last
// NaN of the row-major scan).  -0 and +0 compare equal (`val > maxval`), so -0 is keyed as +0.  0 = "no tap".
__device__ __forceinline__ unsigned long long winner_key(float x, int pos) {
    unsigned u = __float_as_uint(x);
    const bool nan = x != x;
    if (u == 0x80000000u) u = 0u;
    const unsigned ord = nan ? 0xffffffffu : ((u & 0x80000000u) ? ~u : (u | 0x80000000u));
    return ((unsigned long long)ord << 32) | (unsigned)(nan ? pos : 0xffff - pos);
}
template <int KMAX, bool ALONG_X, int SHIFT>
__device__ __forceinline__ void key5_pass(const unsigned long long* src, unsigned long long* dst, unsigned* sel, const Lanes<KMAX>& L, int stride) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int b = L.base[k];
        if (b >= 0) {
            const unsigned m = L.mask[k] >> (ALONG_X ? 0 : 8);
            unsigned long long kk[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const bool ok = (m >> t) & 1u;
                const unsigned long long x = src[ok ? b + (t - 2) * stride : b];
                kk[t] = ok ? x : 0ull;
            }
            unsigned long long best = kk[2]; unsigned mt = 2u;                 // the centre tap is always valid
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                if (t == 2) continue;
                const bool c = kk[t] > best;
                best = c ? kk[t] : best; mt = c ? (unsigned)t : mt;
            }
            dst[b] = best; sel[b] |= mt << SHIFT;
        }
    }
    lds_barrier();
}

// Backward of one 1-D pass: element p collects the gradient of every output q = p + (2 - t) whose recorded winner is tap t.
template <int KMAX, bool ALONG_X, int SHIFT>
__device__ __forceinline__ void back5_pass(const float* gin, float* gout, const unsigned* sel, const Lanes<KMAX>& L, int stride) {
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int b = L.base[k];
        if (b >= 0) {
            const unsigned m = L.mask[k] >> (ALONG_X ? 0 : 8);
            float a = 0.f;
#pragma unroll
            for (int t = 0; t < 5; ++t) {
                const bool ok = (m >> (4 - t)) & 1u;                           // neighbour p + (2 - t) exists
                const int q = ok ? b + (2 - t) * stride : b;
                const unsigned sv = sel[q];
                const float gv = gin[q];
                a += (ok && ((sv >> SHIFT) & 7u) == (unsigned)t) ? gv : 0.0f;
            }
            gout[b] = a;
        }
    }
    lds_barrier();
}

// global <-> LDS tile moves; `e0` = element offset of the tile's first channel, row_stride in elements (both multiples of 4);
// global data is fp32 or bf16 (quads of 16 / 8 bytes), the LDS tile is always fp32
template <int CT, bool BF>
__device__ __forceinline__ void tile_store(const float* lds, void* g, long long e0, long long row_stride, int HW) {
    constexpr int Q = CT / 4;
    for (int i = threadIdx.x; i < HW * Q; i += kTileT) {
        const int pos = i / Q, q = i - pos * Q;
        st_quad<BF>(g, (e0 + pos * row_stride) / 4 + q, *reinterpret_cast<const float4*>(lds + pos * CT + 4 * q));
    }
}
template <int CT, bool BF>
__device__ __forceinline__ void tile_load(float* lds, const void* g, long long e0, long long row_stride, int HW) {
    constexpr int Q = CT / 4;
    for (int i = threadIdx.x; i < HW * Q; i += kTileT) {
        const int pos = i / Q, q = i - pos * Q;
        *reinterpret_cast<float4*>(lds + pos * CT + 4 * q) = ld_quad<BF>(g, (e0 + pos * row_stride) / 4 + q);
    }
}
template <int CT, bool BF>
__device__ __forceinline__ void tile_load_add(float* lds, const void* g, long long e0, long long row_stride, int HW) {
    constexpr int Q = CT / 4;
    for (int i = threadIdx.x; i < HW * Q; i += kTileT) {
        const int pos = i / Q, q = i - pos * Q;
        float4* d = reinterpret_cast<float4*>(lds + pos * CT + 4 * q);
        const float4 a = *d, b = ld_quad<BF>(g, (e0 + pos * row_stride) / 4 + q);
        *d = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
    }
}
template <bool BF>
__device__ __forceinline__ float ld_elem(const void* g, long long e) {
    return BF ? __uint_as_float((unsigned)reinterpret_cast<const unsigned short*>(g)[e] << 16) : reinterpret_cast<const float*>(g)[e];
}
template <bool BF>
__device__ __forceinline__ void st_elem(void* g, long long e, float v) {
    if (BF) reinterpret_cast<unsigned short*>(g)[e] = (unsigned short)bf16_rne(v); else reinterpret_cast<float*>(g)[e] = v;
}

template <int CT, int KMAX, bool BF>
__global__ __launch_bounds__(kTileT) void spp_fwd_nhwc_kernel(const SppParams P) {
    extern __shared__ float lds[];
    const int HW = P.H * P.W;
    const long long C = P.C;
    float* A = lds; float* T = lds + HW * CT;                                     // a stage is row pass A -> T, column pass T -> A (in place)
    const long long x0 = blockIdx.y * P.xs[0] + blockIdx.x * CT, y0 = blockIdx.y * P.ys[0] + blockIdx.x * CT;
    tile_load<CT, BF>(A, P.x, x0, C, HW);
    Lanes<KMAX> L;
    lanes_init<CT, KMAX>(L, P.H, P.W, P.magic_w);
    lds_barrier();
    tile_store<CT, BF>(A, P.y, y0 + P.blk[0] * C, 4 * C, HW);                     // v4: [m13 | m9 | m5 | x]; v5: [x | m5 | m9 | m13]
#pragma unroll
    for (int stage = 1; stage <= 3; ++stage) {
        max5_pass<KMAX, true>(A, T, L, CT);
        max5_pass<KMAX, false>(T, A, L, P.W * CT);
        tile_store<CT, BF>(A, P.y, y0 + P.blk[stage] * C, 4 * C, HW);
    }
}

// Backward along the cascade.  m5 = S(x), m9 = S(m5), m13 = S(m9) with S = (column five-tap max) o (row five-tap max); the forward
// is recomputed with (value, ORIGINAL position) pairs so that every winner is ATen's (first maximum in row-major window order, the
// last NaN), and each of the six 1-D passes records which of its five taps won.  The gradient then walks the same passes in
// reverse -- d m13 through S, joined by d m9, through S, joined by d m5, through S, joined by the identity branch -- five
// (selection, gradient) LDS reads per element and pass: 60 instead of the 550 of a direct gather over the 13 / 9 / 5 windows
// (config 5: 484 -> ~60 us).  The route of a window's gradient ends at the original position its winner carried, so the result is the
// window arg-max scatter of max_pool2d's backward; sums run in a fixed order (bitwise reproducible).
template <int CT, int KMAX, bool BF>
__global__ __launch_bounds__(kTileT) void spp_bwd_nhwc_kernel(const SppParams P) {
    extern __shared__ float lds[];
    const int HW = P.H * P.W, n = HW * CT;
    const long long C = P.C;
    unsigned long long* K0 = reinterpret_cast<unsigned long long*>(lds);       // a stage: row pass K0 -> K1, column pass back into K0
    unsigned long long* K1 = K0 + n;
    unsigned* SEL = reinterpret_cast<unsigned*>(K1 + n);
    float* X = reinterpret_cast<float*>(K1);                                   // the x tile lands in K1's space, keys are built into K0
    const long long x0 = blockIdx.y * P.xs[0] + blockIdx.x * CT, g0 = blockIdx.y * P.ys[0] + blockIdx.x * CT;
    tile_load<CT, BF>(X, P.x, x0, C, HW);
    Lanes<KMAX> L;
    lanes_init<CT, KMAX>(L, P.H, P.W, P.magic_w);
    lds_barrier();
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int b = L.base[k];
        if (b >= 0) { K0[b] = winner_key(X[b], b / CT); SEL[b] = 0u; }
    }
    lds_barrier();
    const int sx = CT, sy = P.W * CT;
    key5_pass<KMAX, true, 0>(K0, K1, SEL, L, sx);   key5_pass<KMAX, false, 3>(K1, K0, SEL, L, sy);      // m5: row winner in bits 0-2, column in 3-5
    key5_pass<KMAX, true, 6>(K0, K1, SEL, L, sx);   key5_pass<KMAX, false, 9>(K1, K0, SEL, L, sy);      // m9
    key5_pass<KMAX, true, 12>(K0, K1, SEL, L, sx);  key5_pass<KMAX, false, 15>(K1, K0, SEL, L, sy);     // m13
    float* Ga = reinterpret_cast<float*>(K0); float* Gb = reinterpret_cast<float*>(K1);                 // the keys are dead: gradient ping-pong
    tile_load<CT, BF>(Ga, P.gy, g0 + P.blk[3] * C, 4 * C, HW);                 // d m13
    lds_barrier();
    back5_pass<KMAX, false, 15>(Ga, Gb, SEL, L, sy);
    back5_pass<KMAX, true, 12>(Gb, Ga, SEL, L, sx);                            // = d m9 through the third stage
    tile_load_add<CT, BF>(Ga, P.gy, g0 + P.blk[2] * C, 4 * C, HW);             // + d m9
    lds_barrier();
    back5_pass<KMAX, false, 9>(Ga, Gb, SEL, L, sy);
    back5_pass<KMAX, true, 6>(Gb, Ga, SEL, L, sx);
    tile_load_add<CT, BF>(Ga, P.gy, g0 + P.blk[1] * C, 4 * C, HW);             // + d m5
    lds_barrier();
    back5_pass<KMAX, false, 3>(Ga, Gb, SEL, L, sy);
    back5_pass<KMAX, true, 0>(Gb, Ga, SEL, L, sx);
    tile_load_add<CT, BF>(Ga, P.gy, g0 + P.blk[0] * C, 4 * C, HW);             // + the identity branch
    lds_barrier();
    tile_store<CT, BF>(Ga, P.gx, x0, C, HW);
}

// dense NHWC on both sides?  -> channel-tile width (32, 16) or 0 for the per-plane kernels
int nhwc_tile(const SppParams& P) {
    const long long C = P.C, W = P.W, H = P.H;
    const bool x_ok = P.xs[1] == 1 && P.xs[3] == C && P.xs[2] == W * C && P.xs[0] == H * W * C;
    const bool y_ok = P.ys[1] == 1 && P.ys[3] == 4 * C && P.ys[2] == W * 4 * C && P.ys[0] == H * W * 4 * C;
    if (!x_ok || !y_ok) return 0;
    if (C % 32 == 0 && H * W <= 176) return 32;
    if (C % 16 == 0 && H * W <= 400) return 16;
    return 0;
}

bool fill(SppParams& P, const float* x, const int64_t* xs, int B, int C, int H, int W, const int64_t* ys) {
    if (!x || !xs || !ys || B < 1 || C < 1 || H < 1 || W < 1 || H * W > kMaxHW || B > 65535) return false;
    P.x = x; P.B = B; P.C = C; P.H = H; P.W = W; P.magic_w = fastdiv16_magic((unsigned)W);
    for (int i = 0; i < 4; ++i) { P.xs[i] = xs[i]; P.ys[i] = ys[i]; }
    return true;
}

template <typename K>
int launch_tile(K kernel, size_t lds_max, size_t lds, dim3 grid, hipStream_t stream, const SppParams& P, bool* attr_done) {
    if (!*attr_done) {          // > 64 KB of dynamic LDS needs the opt-in once per kernel instantiation
        ODPL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_max));
        *attr_done = true;
    }
    hipLaunchKernelGGL(kernel, grid, dim3(kTileT), lds, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

bool set_order(SppParams& P, int order) {
    if (order == ODPL_SPP_ORDER_V4) { P.blk[0] = 3; P.blk[1] = 2; P.blk[2] = 1; P.blk[3] = 0; return true; }
    if (order == ODPL_SPP_ORDER_V5) { P.blk[0] = 0; P.blk[1] = 1; P.blk[2] = 2; P.blk[3] = 3; return true; }
    return false;
}
}  // namespace

extern "C" int odpl_spp_forward_ex(const void* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                   void* y, const int64_t* y_strides, int32_t bf16, int32_t order, void* stream_) {
    SppParams P;
    if (!y || !fill(P, (const float*)x, x_strides, B, C, H, W, y_strides) || !set_order(P, order)) return ODPL_EINVAL;
    P.y = (float*)y; P.gy = nullptr; P.gx = nullptr;
    hipStream_t stream = (hipStream_t)stream_;
    const int ct = (((uintptr_t)x | (uintptr_t)y) & (bf16 ? 7 : 15)) ? 0 : nhwc_tile(P);
    if (ct) {
        const size_t lds = (size_t)2 * H * W * ct * sizeof(float);
        static bool done[4] = {false, false, false, false};
        const dim3 grid(C / ct, B);
        if (ct == 32) return bf16 ? launch_tile(spp_fwd_nhwc_kernel<32, 6, true>, 2 * 176 * 32 * 4, lds, grid, stream, P, &done[0])
                                  : launch_tile(spp_fwd_nhwc_kernel<32, 6, false>, 2 * 176 * 32 * 4, lds, grid, stream, P, &done[1]);
        return bf16 ? launch_tile(spp_fwd_nhwc_kernel<16, 7, true>, 2 * 400 * 16 * 4, lds, grid, stream, P, &done[2])
                    : launch_tile(spp_fwd_nhwc_kernel<16, 7, false>, 2 * 400 * 16 * 4, lds, grid, stream, P, &done[3]);
    }
    if (bf16) return ODPL_EINVAL;                            // bf16 only on the channels-last tile path
    hipLaunchKernelGGL(spp_fwd_kernel, dim3(C, B), dim3(256), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_spp_backward_ex(const void* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                    const void* grad_y, const int64_t* y_strides, void* grad_x, int32_t bf16, int32_t order, void* stream_) {
    SppParams P;
    if (!grad_y || !grad_x || !fill(P, (const float*)x, x_strides, B, C, H, W, y_strides) || !set_order(P, order)) return ODPL_EINVAL;
    P.y = nullptr; P.gy = (const float*)grad_y; P.gx = (float*)grad_x;
    hipStream_t stream = (hipStream_t)stream_;
    const int ct = (((uintptr_t)x | (uintptr_t)grad_y | (uintptr_t)grad_x) & (bf16 ? 7 : 15)) ? 0 : nhwc_tile(P);
    if (ct) {
        const size_t lds = (size_t)H * W * ct * (2 * sizeof(unsigned long long) + sizeof(unsigned));
        static bool done[4] = {false, false, false, false};
        const dim3 grid(C / ct, B);
        if (ct == 32) return bf16 ? launch_tile(spp_bwd_nhwc_kernel<32, 6, true>, 176 * 32 * 20, lds, grid, stream, P, &done[0])
                                  : launch_tile(spp_bwd_nhwc_kernel<32, 6, false>, 176 * 32 * 20, lds, grid, stream, P, &done[1]);
        return bf16 ? launch_tile(spp_bwd_nhwc_kernel<16, 7, true>, 400 * 16 * 20, lds, grid, stream, P, &done[2])
                    : launch_tile(spp_bwd_nhwc_kernel<16, 7, false>, 400 * 16 * 20, lds, grid, stream, P, &done[3]);
    }
    if (bf16) return ODPL_EINVAL;
    hipLaunchKernelGGL(spp_bwd_kernel, dim3(C, B), dim3(256), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_spp_forward(const float* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                float* y, const int64_t* y_strides, void* stream) {
    return odpl_spp_forward_ex(x, x_strides, B, C, H, W, y, y_strides, 0, ODPL_SPP_ORDER_V4, stream);
}

extern "C" int odpl_spp_backward(const float* x, const int64_t* x_strides, int32_t B, int32_t C, int32_t H, int32_t W,
                                 const float* grad_y, const int64_t* y_strides, float* grad_x, void* stream) {
    return odpl_spp_backward_ex(x, x_strides, B, C, H, W, grad_y, y_strides, grad_x, 0, ODPL_SPP_ORDER_V4, stream);
}
