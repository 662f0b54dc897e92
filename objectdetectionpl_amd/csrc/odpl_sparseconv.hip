// Backward of a 3x3 (stride 1, pad 1) convolution whose OUTPUT GRADIENT is zero on most pixels.
//
// RetinaNet's head towers (model/RetinaNet.py:93-99: four Conv2d(256, 256, 3, 1, 1) + ReLU per head, shared by the five FPN levels) sit
// behind the two final convolutions, whose input gradient is non-zero only within one pixel of a positive anchor (csrc/odpl_retinagrad.hip).
// Each 3x3 layer further back widens that support by one pixel: on the two large levels (94 % of the towers' arithmetic) a few per cent
// of the pixels carry any gradient at all, yet a library convolution backward multiplies all of them -- 16 ms of the BASELINE config-4
// step.  Here the two backward products run over the ACTIVE pixels only:
//     dX[q, :]        = sum_tap W_tap^T . gY[q - d(tap), :]          for q within one pixel of an active pixel; zero elsewhere
//     dW[:, :, tap]   = sum_{p active} gY[p, :] (x) X[p + d(tap), :]
// as gathered GEMMs on the fp32 matrix cores (v_mfma_f32_32x32x2_f32, the tiling of csrc/odpl_gemm1x1.hip: 64 x 64 tiles, K step 32,
// padded LDS images, next tile prefetched into registers).  Which pixels are active is read from a byte map the preceding bias + ReLU
// backward launch emits for free (odpl_bias_act_backward_flags).  Launches: tile counts -> scan -> ordered lists (ascending pixel: the
// sums below run in list order, no float atomics, bitwise reproducible) -> zero fill of dX -> data GEMM -> weight GEMM (split over the
// list, partials) -> fixed-order reduction.  Nothing is read back by the host: grids are sized for the worst case and walk device-side counts.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT = 256;
constexpr int BK = 32;
constexpr int kSplits = 8;              // pixel-list chunks of the weight GEMM (partials reduced in chunk order)

struct SP {
    const float* x; const float* gy; const unsigned char* act; const float* w; float* gx; float* gw;
    long long ws_o, ws_i, ws_ky, ws_kx;
    int B, H, W, Cin, Cout, P, ntile;    // P = B*H*W pixels; ntile = ceil(P / 256)
    int ks, pad, taps;                   // 3 x 3 (pad 1) or 1 x 1 (pad 0): taps = ks * ks
    int* cnt_out; int* cnt_in;           // [ntile] active pixels / pixels within one pixel of an active one, per 256-pixel tile
    int* off_out; int* off_in;           // [ntile + 1] exclusive scans; [ntile] = totals
    int* list_out; int* list_in;         // [P] ascending pixel ids
    float* part;                         // [kSplits][9][Cout][Cin]
    float* part_x; long long part_x_floats;   // split-K partials of the data GEMM on short lists: [R][tiles_m * 64][Cin]
    int* stats;                          // optional [2]: active pixels, pixels within one pixel of an active one (for the caller's route choice)
};

__device__ __forceinline__ bool near_active(const SP& P, int p) {
    const int hw = P.H * P.W, b = p / hw, r = p - b * hw, y = r / P.W, x = r - y * P.W;
    bool any = false;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            if ((dy != 0 || dx != 0) && P.pad == 0) continue;             // a 1 x 1 convolution reaches no neighbour
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < P.H && xx >= 0 && xx < P.W) any |= P.act[(size_t)b * hw + yy * P.W + xx] != 0;
        }
    return any;
}

__global__ __launch_bounds__(kT) void sc_count_kernel(const SP P) {
    __shared__ int s_w[2][kT / 64];
    const int p = blockIdx.x * kT + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = p < P.P;
    const bool o = live && P.act[p] != 0, i = live && near_active(P, p);
    const unsigned long long mo = __ballot(o), mi = __ballot(i);
    if (lane == 0) { s_w[0][wv] = __popcll(mo); s_w[1][wv] = __popcll(mi); }
    __syncthreads();
    if (threadIdx.x == 0) {
        int a = 0, c = 0;
        for (int w = 0; w < kT / 64; ++w) { a += s_w[0][w]; c += s_w[1][w]; }
        P.cnt_out[blockIdx.x] = a; P.cnt_in[blockIdx.x] = c;
    }
}

__global__ __launch_bounds__(1024) void sc_scan_kernel(const SP P) {
    __shared__ int s_w[2][16];
    __shared__ int s_carry[2];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid < 2) s_carry[tid] = 0;
    __syncthreads();
    for (int base = 0; base < P.ntile; base += 1024) {
        const int i = base + tid;
        const int c0 = i < P.ntile ? P.cnt_out[i] : 0, c1 = i < P.ntile ? P.cnt_in[i] : 0;
        int v0 = c0, v1 = c1;
        for (int o = 1; o < 64; o <<= 1) { const int u0 = __shfl_up(v0, o, 64), u1 = __shfl_up(v1, o, 64); if (lane >= o) { v0 += u0; v1 += u1; } }
        if (lane == 63) { s_w[0][wv] = v0; s_w[1][wv] = v1; }
        __syncthreads();
        int w0 = 0, w1 = 0;
        for (int w = 0; w < wv; ++w) { w0 += s_w[0][w]; w1 += s_w[1][w]; }
        const int e0 = s_carry[0] + w0 + v0 - c0, e1 = s_carry[1] + w1 + v1 - c1;
        if (i < P.ntile) { P.off_out[i] = e0; P.off_in[i] = e1; }
        __syncthreads();
        if (tid == 1023) { s_carry[0] = e0 + c0; s_carry[1] = e1 + c1; }
        __syncthreads();
    }
    if (tid == 0) {
        P.off_out[P.ntile] = s_carry[0]; P.off_in[P.ntile] = s_carry[1];
        if (P.stats) { P.stats[0] = s_carry[0]; P.stats[1] = s_carry[1]; }
    }
}

__global__ __launch_bounds__(kT) void sc_list_kernel(const SP P) {
    __shared__ int s_w[2][kT / 64];
    const int p = blockIdx.x * kT + threadIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const bool live = p < P.P;
    const bool o = live && P.act[p] != 0, i = live && near_active(P, p);
    const unsigned long long mo = __ballot(o), mi = __ballot(i);
    if (lane == 0) { s_w[0][wv] = __popcll(mo); s_w[1][wv] = __popcll(mi); }
    __syncthreads();
    int b0 = P.off_out[blockIdx.x], b1 = P.off_in[blockIdx.x];
    for (int w = 0; w < wv; ++w) { b0 += s_w[0][w]; b1 += s_w[1][w]; }
    const unsigned long long below = (1ull << lane) - 1ull;
    if (o) P.list_out[b0 + __popcll(mo & below)] = p;
    if (i) P.list_in[b1 + __popcll(mi & below)] = p;
}

// K (= taps * Cout / 32 tiles) is ALWAYS summed as R = 8 consecutive ranges whose sums are added left to right; a short list (a few dozen
// output tiles, each a serial walk over all of K at the matrix pipe's pace) gives every range its own workgroup and sc_data_finish_kernel
// adds the partials in the same order: same bits whichever way a launch went (csrc/odpl_sparsefwd.hip does the same going forward).
__device__ __forceinline__ int k_ranges(int nk) { return nk < 8 ? nk : 8; }
__device__ __forceinline__ int k_bound(int r, int nk, int R) { return (int)((long long)r * nk / R); }
__device__ __forceinline__ bool split_ranges(int ntiles, int tiles_m, int cin, int R, const float* part, long long part_floats) {
    return part && R > 1 && ntiles * R <= 1024 && (long long)tiles_m * 64 * cin * R <= part_floats;
}

// dX rows of the pixels in list_in: [n_in, taps * Cout] (gathered rows of gY, one tap at a time) x [taps * Cout, Cin]
__global__ __launch_bounds__(kT) void sc_data_kernel(const SP P) {
    constexpr int BM = 64, BN = 64, LDA = BK + 1, LDB = BN + 4;
    __shared__ float As[BM * LDA];
    __shared__ float Bs[BK * LDB];
    __shared__ int s_b[BM], s_y[BM], s_x[BM], s_q[BM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32, lr = lane & 31, lk = lane >> 5;
    const int n_in = P.off_in[P.ntile];
    const int tiles_n = P.Cin / BN, tiles_m = (n_in + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int hw = P.H * P.W, kpt = P.Cout / BK;                         // K steps per tap
    const int nk_all = P.taps * kpt, R = k_ranges(nk_all);
    const int S = split_ranges(ntiles, tiles_m, P.Cin, R, P.part_x, P.part_x_floats) ? R : 1;
    for (int work = blockIdx.x; work < ntiles * S; work += gridDim.x) {
        const int tile = work / S, sp = work - tile * S;
        const int k_begin = S == 1 ? 0 : k_bound(sp, nk_all, R), k_end = S == 1 ? nk_all : k_bound(sp + 1, nk_all, R);
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n, bn = tn * BN;
        __syncthreads();
        if (tid < BM) {
            int r = tm * BM + tid; if (r > n_in - 1) r = n_in - 1;       // clamped: the surplus rows are computed and not stored
            const int q = P.list_in[r], b = q / hw, rem = q - b * hw, y = rem / P.W;
            s_q[tid] = q; s_b[tid] = b; s_y[tid] = y; s_x[tid] = rem - y * P.W;
        }
        __syncthreads();
        f32x16 acc, total;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; total[r] = 0.f; }
        int cur = 0, next_bound = S == 1 ? k_bound(1, nk_all, R) : k_end;
        float4 ra[2], rb[2];
        auto load_tiles = [&](int kt) {
            const int t = kt / kpt, co0 = (kt - t * kpt) * BK, ky = t / P.ks, kx = t - P.ks * ky;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, row = f >> 3, kq = f & 7;
                const int yy = s_y[row] - (ky - P.pad), xx = s_x[row] - (kx - P.pad);        // dX[q] takes gY[q - d(tap)]
                const bool in = yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
                const size_t src = ((size_t)s_b[row] * hw + (in ? yy * P.W + xx : 0)) * P.Cout + co0 + 4 * kq;
                const float4 v = *reinterpret_cast<const float4*>(P.gy + src);
                ra[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, kr = f >> 4, nq = f & 15;                        // W[co0 + kr][bn + 4 nq ..][ky][kx]
                const long long off = (long long)(co0 + kr) * P.ws_o + ky * P.ws_ky + kx * P.ws_kx + (long long)(bn + 4 * nq) * P.ws_i;
                rb[i] = P.ws_i == 1 ? *reinterpret_cast<const float4*>(P.w + off)
                                    : make_float4(P.w[off], P.w[off + P.ws_i], P.w[off + 2 * P.ws_i], P.w[off + 3 * P.ws_i]);
            }
        };
        auto store_tiles = [&]() {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, row = f >> 3, kq = f & 7;
                float* d = As + row * LDA + 4 * kq;
                d[0] = ra[i].x; d[1] = ra[i].y; d[2] = ra[i].z; d[3] = ra[i].w;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, kr = f >> 4, nq = f & 15;
                *reinterpret_cast<float4*>(Bs + kr * LDB + 4 * nq) = rb[i];
            }
        };
        load_tiles(k_begin);
        store_tiles();
        __syncthreads();
        for (int kt = k_begin; kt < k_end; ++kt) {
            if (kt + 1 < k_end) load_tiles(kt + 1);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a = As[(wm + lr) * LDA + kk + lk], b = Bs[(kk + lk) * LDB + wn + lr];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            if (S == 1 && kt + 1 == next_bound) {                        // unsplit: fold the range sums in order as they complete
#pragma unroll
                for (int r = 0; r < 16; ++r) { total[r] += acc[r]; acc[r] = 0.f; }
                ++cur; next_bound = k_bound(cur + 1, nk_all, R);
            }
            __syncthreads();
            if (kt + 1 < k_end) { store_tiles(); __syncthreads(); }
        }
        if (S > 1) {
            float* o = P.part_x + ((size_t)sp * tiles_m * BM + (size_t)tm * BM) * P.Cin + bn + wn + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)(wm + (r & 3) + 8 * (r >> 2) + 4 * lk) * P.Cin] = acc[r];
            continue;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (tm * BM + row < n_in) P.gx[(size_t)s_q[row] * P.Cin + bn + wn + lr] = total[r];
        }
    }
}

// the split launches' second half: dX[q, ci] = sum over the R ranges in range order; returns at once when the launch was not split
__global__ __launch_bounds__(kT) void sc_data_finish_kernel(const SP P) {
    const int n_in = P.off_in[P.ntile];
    const int tiles_n = P.Cin / 64, tiles_m = (n_in + 63) / 64, kpt = P.Cout / BK;
    const int R = k_ranges(P.taps * kpt);
    if (!split_ranges(tiles_m * tiles_n, tiles_m, P.Cin, R, P.part_x, P.part_x_floats)) return;
    const size_t plane = (size_t)tiles_m * 64 * P.Cin;
    const int c4n = P.Cin >> 2;
    const long long total = (long long)n_in * c4n;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
        const int r = (int)(i / c4n), c = (int)(i - (long long)r * c4n);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);                      // 0 + r0 + r1 + ...: the order of the unsplit launch
        for (int s2 = 0; s2 < R; ++s2) {
            const float4 p = *reinterpret_cast<const float4*>(P.part_x + s2 * plane + (size_t)r * P.Cin + 4 * c);
            v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        *reinterpret_cast<float4*>(P.gx + (size_t)P.list_in[r] * P.Cin + 4 * c) = v;
    }
}

// partial dW of one tap over one chunk of list_out: [Cout, n] (gathered rows of gY) x [n, Cin] (rows of X at p + d(tap))
__global__ __launch_bounds__(kT) void sc_weight_kernel(const SP P) {
    constexpr int BM = 64, BN = 64, LD = 64 + 4;
    __shared__ float As[BK * LD];                                        // [pixel][co]
    __shared__ float Bs[BK * LD];                                        // [pixel][ci]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32, lr = lane & 31, lk = lane >> 5;
    const int tiles_m = P.Cout / BM, tiles_n = P.Cin / BN;
    int bid = blockIdx.x;
    const int tn = bid % tiles_n; bid /= tiles_n;
    const int tm = bid % tiles_m; bid /= tiles_m;
    const int t = bid % P.taps, s = bid / P.taps;
    const int ky = t / P.ks, kx = t - P.ks * ky, hw = P.H * P.W;
    const int n_out = P.off_out[P.ntile];
    const int chunk = ((n_out + kSplits - 1) / kSplits + BK - 1) / BK * BK;
    const int r0 = s * chunk, r1 = r0 + chunk < n_out ? r0 + chunk : n_out;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float4 ra[2], rb[2];
    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + kT * i, kr = f >> 4, cq = f & 15;
            const int r = k0 + kr;
            const bool live = r < r1;
            const int p = P.list_out[live ? r : r0 < n_out ? r0 : 0];
            const float4 g = *reinterpret_cast<const float4*>(P.gy + (size_t)p * P.Cout + tm * BM + 4 * cq);
            ra[i] = live ? g : make_float4(0.f, 0.f, 0.f, 0.f);
            const int b = p / hw, rem = p - b * hw, y = rem / P.W, x = rem - y * P.W, yy = y + ky - P.pad, xx = x + kx - P.pad;
            const bool in = live && yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
            const float4 v = *reinterpret_cast<const float4*>(P.x + ((size_t)b * hw + (in ? yy * P.W + xx : rem)) * P.Cin + tn * BN + 4 * cq);
            rb[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int f = tid + kT * i, kr = f >> 4, cq = f & 15;
            *reinterpret_cast<float4*>(As + kr * LD + 4 * cq) = ra[i];
            *reinterpret_cast<float4*>(Bs + kr * LD + 4 * cq) = rb[i];
        }
    };
    if (r0 < r1) {
        load_tiles(r0);
        store_tiles();
        __syncthreads();
        for (int k0 = r0; k0 < r1; k0 += BK) {
            if (k0 + BK < r1) load_tiles(k0 + BK);
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                const float a = As[(kk + lk) * LD + wm + lr], b = Bs[(kk + lk) * LD + wn + lr];
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
            }
            __syncthreads();
            if (k0 + BK < r1) { store_tiles(); __syncthreads(); }
        }
    }
    float* o = P.part + ((size_t)(s * P.taps + t) * P.Cout + tm * BM) * P.Cin + tn * BN;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[(size_t)(wm + (r & 3) + 8 * (r >> 2) + 4 * lk) * P.Cin + wn + lr] = acc[r];
}

__global__ __launch_bounds__(kT) void sc_reduce_kernel(const SP P) {
    const long long n = (long long)P.taps * P.Cout * P.Cin;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < n; i += (long long)gridDim.x * kT) {
        float v = P.part[i];
        for (int s = 1; s < kSplits; ++s) v += P.part[(long long)s * n + i];
        const int ci = (int)(i % P.Cin); const long long r = i / P.Cin; const int co = (int)(r % P.Cout), t = (int)(r / P.Cout);
        P.gw[(long long)co * P.ws_o + (long long)ci * P.ws_i + (t / P.ks) * P.ws_ky + (t % P.ks) * P.ws_kx] = v;
    }
}

// ---- pixel sets of a coarser FPN level: what a finer level's top-down merge will read of it ---------------------------------
struct UnP {
    unsigned char* flags;
    const int* list_a; const int* count_a;       // pixels of THIS map, taken as they are (may be null)
    const int* list_b; const int* count_b;       // pixels of a FINER map (Hf x Wf): their bilinear source pixels in this map (may be null)
    int B, H, W, Hf, Wf;
    float sy, sx;
};

__global__ __launch_bounds__(kT) void un_mark_kernel(const UnP P) {
    const int stride = gridDim.x * kT, t0 = blockIdx.x * kT + threadIdx.x;
    if (P.list_a) {
        const int n = min(P.count_a[0], P.B * P.H * P.W);
        for (int r = t0; r < n; r += stride) P.flags[P.list_a[r]] = 1;
    }
    if (P.list_b) {
        const int n = min(P.count_b[0], P.B * P.Hf * P.Wf), hwf = P.Hf * P.Wf;
        for (int r = t0; r < n; r += stride) {
            const int q = P.list_b[r], b = q / hwf, rem = q - b * hwf, y = rem / P.Wf, x = rem - y * P.Wf;
            float fy = P.sy * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;      // the source index arithmetic of sf_upadd_kernel / ATen
            float fx = P.sx * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
            const int y0 = (int)fy, x0 = (int)fx;
            const int y1 = y0 + (y0 < P.H - 1 ? 1 : 0), x1 = x0 + (x0 < P.W - 1 ? 1 : 0);
            unsigned char* f = P.flags + (size_t)b * P.H * P.W;
            f[y0 * P.W + x0] = 1; f[y0 * P.W + x1] = 1; f[y1 * P.W + x0] = 1; f[y1 * P.W + x1] = 1;
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct Plan { size_t cnt_out, cnt_in, off_out, off_in, list_out, list_in, part, part_x, total; int P, ntile; };
constexpr size_t kPartXBytes = 32u << 20;      // split-K partials of the data GEMM (short lists only)

bool make_plan(const odpl_sparse_conv3x3_t* d, Plan* p) {
    if (!d || d->B < 1 || d->H < 1 || d->W < 1 || d->Cin < 64 || (d->Cin % 64) || d->Cout < 64 || (d->Cout % 64)) return false;
    const long long P = (long long)d->B * d->H * d->W;
    if (P * d->Cin > 0x7fffffffLL * 4 || P * d->Cout > 0x7fffffffLL * 4 || P > 0x7fffffffLL - 256) return false;
    if (!d->x || !d->gy || !d->active || !d->w) return false;
    if (d->ksize != 0 && d->ksize != 1 && d->ksize != 3) return false;
    p->P = (int)P; p->ntile = (int)((P + kT - 1) / kT);
    size_t off = 0;
    p->cnt_out = off; off += align_up((size_t)p->ntile * 4, 256);
    p->cnt_in = off; off += align_up((size_t)p->ntile * 4, 256);
    p->off_out = off; off += align_up(((size_t)p->ntile + 1) * 4, 256);
    p->off_in = off; off += align_up(((size_t)p->ntile + 1) * 4, 256);
    p->list_out = off; off += align_up((size_t)P * 4, 256);
    p->list_in = off; off += align_up((size_t)P * 4, 256);
    p->part = off; off += align_up((size_t)kSplits * (d->ksize == 1 ? 1 : 9) * d->Cout * d->Cin * 4, 256);
    p->part_x = off; off += kPartXBytes;
    p->total = off;
    return true;
}
}  // namespace

extern "C" size_t odpl_sparse_conv3x3_backward_workspace_bytes(const odpl_sparse_conv3x3_t* d) {
    Plan p;
    return make_plan(d, &p) ? p.total : 0;     // (host only: reads no device memory)
}

extern "C" int odpl_sparse_conv3x3_backward(const odpl_sparse_conv3x3_t* d, void* workspace, size_t workspace_bytes, void* stream_) {
    Plan pl;
    if (!make_plan(d, &pl) || !workspace || workspace_bytes < pl.total) return ODPL_EINVAL;
    if ((((uintptr_t)workspace | (uintptr_t)d->x | (uintptr_t)d->gy | (uintptr_t)d->grad_x) & 15)) return ODPL_EINVAL;
    if (d->w_stride[1] == 1 && ((((uintptr_t)d->w) & 15) || (d->w_stride[0] & 3) ||
                                (d->ksize != 1 && ((d->w_stride[2] & 3) || (d->w_stride[3] & 3))))) return ODPL_EINVAL;   // (a 1 x 1 kernel never steps along ky / kx)
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    SP P;
    P.x = d->x; P.gy = d->gy; P.act = d->active; P.w = d->w; P.gx = d->grad_x; P.gw = d->grad_w;
    P.ws_o = d->w_stride[0]; P.ws_i = d->w_stride[1]; P.ws_ky = d->w_stride[2]; P.ws_kx = d->w_stride[3];
    P.B = d->B; P.H = d->H; P.W = d->W; P.Cin = d->Cin; P.Cout = d->Cout; P.P = pl.P; P.ntile = pl.ntile;
    P.ks = d->ksize == 1 ? 1 : 3; P.pad = P.ks / 2; P.taps = P.ks * P.ks;
    P.cnt_out = (int*)(ws + pl.cnt_out); P.cnt_in = (int*)(ws + pl.cnt_in); P.off_out = (int*)(ws + pl.off_out); P.off_in = (int*)(ws + pl.off_in);
    P.list_out = (int*)(ws + pl.list_out); P.list_in = (int*)(ws + pl.list_in); P.part = (float*)(ws + pl.part);
    P.stats = d->stats;
    P.part_x = (float*)(ws + pl.part_x); P.part_x_floats = (long long)(kPartXBytes / sizeof(float));
    hipLaunchKernelGGL(sc_count_kernel, dim3(pl.ntile), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_scan_kernel, dim3(1), dim3(1024), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_list_kernel, dim3(pl.ntile), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    if (d->grad_x) {
        ODPL_HIP_TRY(odpl::fill_bytes(d->grad_x, 0, (size_t)pl.P * d->Cin * sizeof(float), stream));
        long long worst = ((long long)pl.P + 63) / 64 * (d->Cin / 64);
        hipLaunchKernelGGL(sc_data_kernel, dim3((unsigned)(worst < 2048 ? worst : 2048)), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(sc_data_finish_kernel, dim3(512), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    if (d->grad_w) {
        hipLaunchKernelGGL(sc_weight_kernel, dim3(kSplits * P.taps * (d->Cout / 64) * (d->Cin / 64)), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(sc_reduce_kernel, dim3(1024), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

// flags | 4 x (ntile [+ 1]) ints of counts / offsets; the two lists are the caller's
static size_t union_plan(long long P, size_t* cnt_out, size_t* cnt_in, size_t* off_out, size_t* off_in) {
    const size_t ntile = (size_t)((P + kT - 1) / kT);
    size_t off = align_up((size_t)P, 256);
    *cnt_out = off; off += align_up(ntile * 4, 256);
    *cnt_in = off; off += align_up(ntile * 4, 256);
    *off_out = off; off += align_up((ntile + 1) * 4, 256);
    *off_in = off; off += align_up((ntile + 1) * 4, 256);
    return off;
}

extern "C" size_t odpl_pixel_union_lists_workspace_bytes(int32_t B, int32_t H, int32_t W) {
    if (B < 1 || H < 1 || W < 1 || (long long)B * H * W > 0x7fffffffLL - 256) return 0;
    size_t a, b, c, d;
    return union_plan((long long)B * H * W, &a, &b, &c, &d);
}

extern "C" int odpl_pixel_union_lists(const int32_t* list_a, const int32_t* count_a, const int32_t* list_b, const int32_t* count_b,
                                      int32_t B, int32_t H, int32_t W, int32_t Hf, int32_t Wf,
                                      int32_t* list_marked, int32_t* list_near, int32_t* counts2,
                                      void* workspace, size_t workspace_bytes, void* stream_) {
    const size_t need = odpl_pixel_union_lists_workspace_bytes(B, H, W);
    if (!need || !workspace || workspace_bytes < need || !list_marked || !list_near || !counts2) return ODPL_EINVAL;
    if ((list_a == nullptr) != (count_a == nullptr) || (list_b == nullptr) != (count_b == nullptr) || (!list_a && !list_b)) return ODPL_EINVAL;
    if (list_b && (Hf < H || Wf < W || (long long)B * Hf * Wf > 0x7fffffffLL)) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    const long long Pn = (long long)B * H * W;
    size_t o_cnt_out, o_cnt_in, o_off_out, o_off_in;
    union_plan(Pn, &o_cnt_out, &o_cnt_in, &o_off_out, &o_off_in);
    UnP U;
    U.flags = (unsigned char*)ws; U.list_a = list_a; U.count_a = count_a; U.list_b = list_b; U.count_b = count_b;
    U.B = B; U.H = H; U.W = W; U.Hf = list_b ? Hf : 1; U.Wf = list_b ? Wf : 1;
    U.sy = list_b ? (float)H / (float)Hf : 1.f; U.sx = list_b ? (float)W / (float)Wf : 1.f;
    ODPL_HIP_TRY(odpl::fill_bytes(U.flags, 0, (size_t)Pn, stream));
    hipLaunchKernelGGL(un_mark_kernel, dim3(256), dim3(kT), 0, stream, U);
    ODPL_HIP_TRY(hipGetLastError());
    SP P = {};
    P.act = U.flags; P.B = B; P.H = H; P.W = W; P.P = (int)Pn; P.ntile = (int)((Pn + kT - 1) / kT);
    P.ks = 3; P.pad = 1; P.taps = 9;
    P.cnt_out = (int*)(ws + o_cnt_out); P.cnt_in = (int*)(ws + o_cnt_in); P.off_out = (int*)(ws + o_off_out); P.off_in = (int*)(ws + o_off_in);
    P.list_out = list_marked; P.list_in = list_near; P.stats = counts2;
    hipLaunchKernelGGL(sc_count_kernel, dim3(P.ntile), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_scan_kernel, dim3(1), dim3(1024), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sc_list_kernel, dim3(P.ntile), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
