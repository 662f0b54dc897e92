// Backward of a detection head's 1x1 convolution from the STRUCTURED gradient the YOLO losses produce.
//
// The linear head convolutions (model/YOLOV4.py:89-127: Conv2d(Cin, A*(5+C), 1) + bias; YOLOV5.py:93-97 Detect.m) feed the
// criterion directly.  d loss / d head is zero except for one channel per (cell, anchor) -- the confidence / objectness
// logit -- and the 4+C other channels of the <= nT matched cells (losses.py:557-564).  A generic convolution backward
// would stream a dense [B, A*(5+C), G, G] gradient that is > 98 % zeros through two GEMMs (57.9 MB and 20 GFLOP per step at
// BASELINE config 2, 0.89 ms measured with MIOpen); here that tensor is never written:
//
//   dX[n, :]  = sum_a gd[n,a] * W[a*no+dc, :]                 (every row n = (b, s): A FMAs per element, a pure write stream)
//             + sum_{rows t at n} sum_{c != dc} gr[t,c] * W[a_t*no+c, :]                        (matched cells only)
//   dW[ch, :] = sum_n gd[n,a] * X[n, :]   for ch = a*no+dc     (a pure read stream of X, A accumulators per column)
//             = sum_{rows t with a_t = a} gr[t,c] * X[n_t, :]  for the other channels           (matched cells only)
//   db[ch]    = the same sums without X
//
// X is the convolution's input in channels-last memory ([N = B*G2, Cin] row-major, fp32 or bf16); gd / gr / row_cell are
// the `odpl_sparse_grad_t` of include/odpl_hip.h.  Both streams are HBM bound (X read once, dX written once): this is the
// kernel the training step's loss-path roofline figure is quoted on.  All heads of a model go through ONE pair of launches.
//   stream  : workgroup = 32 rows x 64 16-byte vectors (64 KB of traffic: 2368 workgroups at config 2, heaviest head
//             first); lanes along Cin (coalesced 1-KiB wave accesses); gd of the tile staged in LDS; eight rows of X in flight
//             per lane; per-tile dW / db partials (no float atomics).  X is loaded non-temporally (dead after this kernel),
//             dX is written with PLAIN stores: on this chip a non-temporal store stream beside a read stream costs a copy
//             17 % (tools/probe_copy.hip: 6.08 vs 5.09 TB/s at this size), and the next backward convolution reads dX
//   finish  : (a) 64 x 4-lane groups per workgroup reduce the partials of the dense channels in a fixed order;
//             (b) one workgroup per other output channel: the matched rows of its anchor are compacted into LDS in row
//             order, then summed four X rows at a time; (c) one workgroup per matched cell-row REWRITES that dX row: the dense
//             part again (the stream kernel's arithmetic) + the matched part, summed in fp32 and stored once, so a bf16 dX is
//             rounded once (rows that share a cell-row are applied by the first of them, in row order).
//             (b) and (c) walk row lists longer than the LDS lists (4096 rows) in chunks: no limit on the number of rows
// Bitwise reproducible run to run.
#include "odpl_common.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

constexpr int kThreads = 256;
constexpr int kMaxA = 5;
constexpr int kRowChunk = 4096;         // matched rows the finish kernel stages in LDS at a time (longer row lists are walked in chunks)
int g_rows_per_tile = 0;               // > 0: a fixed tile height for 64-vector rows (round 2-3: 32); 0 = [r4] equal shares: every workgroup gets
                                       // ~1 / g_share_wgs of the launch's bytes as ONE tall tile and walks it in double-buffered batches
int g_share_wgs = 512;                 // workgroups of the stream launch in equal-share mode (two per CU)

struct Head {
    const void* x; const float* w; void* gx; float* gw; float* gb;
    const float* gd; const float* gr; const int* row_cell; const float* upstream;
    float* part_w;        // [ntiles, A, Cin]
    float* part_b;        // [ntiles, A]
    int N, Cin, cv, cq, G2, A, no, dc, layout, n_rows;
    int ntiles, rows_per_tile, tpr, rows_par, ncs;
    int st_blk0, st_nblk;                 // stream kernel blocks
    int fd_blk0, fd_nblk, fd_vg;          // finish: dense-channel reduce blocks (fd_vg 4-float4 groups per anchor)
    int fc_blk0, fc_nblk;                 // finish: one block per other output channel
    int fr_blk0, fr_nblk;                 // finish: one block per matched row
};
struct Multi { Head h[ODPL_MAX_SCALES]; int n, cap; };      // cap: LDS row-list capacity of the finish kernel (rows per chunk)

template <bool BF> struct Vec { static constexpr int N = BF ? 8 : 4; float v[BF ? 8 : 4]; };

template <bool BF, bool NT = false>
__device__ __forceinline__ Vec<BF> ldv(const void* p, size_t i) {
    Vec<BF> r;
    if (!BF) {
        v4f t4;
        if (NT) t4 = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p) + i); else t4 = reinterpret_cast<const v4f*>(p)[i];
        const float4 t = make_float4(t4[0], t4[1], t4[2], t4[3]);
        r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
    } else {
        typedef unsigned u4v __attribute__((ext_vector_type(4)));
        u4v u;
        if (NT) u = __builtin_nontemporal_load(reinterpret_cast<const u4v*>(p) + i); else u = reinterpret_cast<const u4v*>(p)[i];
        const unsigned w[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) { r.v[2 * k] = __uint_as_float(w[k] << 16); r.v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
    return r;
}
template <bool BF, bool NT>
__device__ __forceinline__ void stv(void* p, size_t i, const Vec<BF>& r) {
    if (!BF) {
        v4f o = {r.v[0], r.v[1], r.v[2], r.v[3]};
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(p) + i); else reinterpret_cast<v4f*>(p)[i] = o;
    } else {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = bf16_rne(r.v[2 * k]) | (bf16_rne(r.v[2 * k + 1]) << 16);
        if (NT) __builtin_nontemporal_store(o, reinterpret_cast<u4*>(p) + i); else reinterpret_cast<u4*>(p)[i] = o;
    }
}
// EPV consecutive floats of an fp32 row (weights, partials)
template <bool BF>
__device__ __forceinline__ Vec<BF> ldf(const float* row, int vec) {
    Vec<BF> r;
    const float4* p = reinterpret_cast<const float4*>(row) + (size_t)vec * (Vec<BF>::N / 4);
#pragma unroll
    for (int k = 0; k < Vec<BF>::N / 4; ++k) { const float4 t = p[k]; r.v[4 * k] = t.x; r.v[4 * k + 1] = t.y; r.v[4 * k + 2] = t.z; r.v[4 * k + 3] = t.w; }
    return r;
}
template <bool BF>
__device__ __forceinline__ void stf(float* row, int vec, const Vec<BF>& r) {
    float4* p = reinterpret_cast<float4*>(row) + (size_t)vec * (Vec<BF>::N / 4);
#pragma unroll
    for (int k = 0; k < Vec<BF>::N / 4; ++k) p[k] = make_float4(r.v[4 * k], r.v[4 * k + 1], r.v[4 * k + 2], r.v[4 * k + 3]);
}

// index of (row n, anchor a) in the dense part
__device__ __forceinline__ size_t gd_index(const Head& P, int n, int a) {
    if (P.layout == ODPL_LAYOUT_PLANAR) { const int b = n / P.G2, s = n - b * P.G2; return ((size_t)b * P.A + a) * P.G2 + s; }
    return (size_t)n * P.A + a;
}
// (row n, anchor a) of a linear cell; n = -1 for "no row" or an out-of-range cell
__device__ __forceinline__ void cell_decode(const Head& P, int cell, int& n, int& a) {
    if (cell < 0) { n = -1; a = 0; return; }
    if (P.layout == ODPL_LAYOUT_PLANAR) { const int plane = cell / P.G2, s = cell - plane * P.G2; a = plane % P.A; n = (plane / P.A) * P.G2 + s; }
    else { a = cell % P.A; n = cell / P.A; }
    if (n >= P.N) n = -1;
}

constexpr int kLaneRows = 8;            // rows of X in flight per lane (8 x 16 B)

template <bool BF, int A>
__global__ __launch_bounds__(kThreads) void headgrad_stream_kernel(const Multi M) {
    constexpr int E = Vec<BF>::N;
    extern __shared__ __align__(16) float lds[];                       // [rows_per_tile * A] gd of the tile, then [rows_par * A * tpr * E] reduction
    const int tid = threadIdx.x, bid = blockIdx.x;
    int hi = 0;
    for (int i = 1; i < M.n; ++i) if (bid >= M.h[i].st_blk0) hi = i;
    const Head& P = M.h[hi];
    const int local = bid - P.st_blk0;
    const int tile = local / P.ncs, cs = local - tile * P.ncs;
    const int n0 = tile * P.rows_per_tile;
    const int nrows = min(P.rows_per_tile, P.N - n0);
    const float f = P.upstream ? *P.upstream : 1.0f;
    float* s_g = lds;
    const int tpr = P.tpr;                               // threads along Cin (power of two, <= 64)
    const int lane_c = tid & (tpr - 1), rl = tid / tpr;  // column slot, row lane
    const int rows_par = P.rows_par;                     // = kThreads / tpr row lanes
    const int cv = P.cv;                                 // 16-byte vectors per row of X
    const int q = cs * tpr + lane_c;
    const bool col_ok = q < cv;
    const int qc = col_ok ? q : cv - 1;                  // loads are unconditional (clamped): a predicated load ends in register
    const bool want_w = P.gw != nullptr, want_x = P.gx != nullptr;     // copies that the compiler must wait for on the spot
    // Addresses: a uniform 64-bit tile base (scalar registers) + a 32-bit byte offset per lane (a tile is < 2^31 bytes).
    const char* tx = reinterpret_cast<const char*>(P.x) + (size_t)n0 * cv * 16;
    char* tg = reinterpret_cast<char*>(P.gx) + (size_t)n0 * cv * 16;
    const unsigned rowb = (unsigned)cv * 16u;
    // Issue order = retirement order.  (1) the tile's gradient values, (2) the A weight rows, (3) kLaneRows rows of X per lane.
    // The staging of (1) into LDS then waits for (1) only, the dX arithmetic for (2) only; X is first needed by the dW sums.
    const int ng = nrows * A;
    const int i0 = tid < ng ? tid : ng - 1;
    const float g0 = P.gd[gd_index(P, n0 + i0 / A, i0 % A)];
    __builtin_amdgcn_sched_barrier(0);                   // keep the instruction scheduler from sinking this load below the X loads
    Vec<BF> wv[A], acc[A];
#pragma unroll
    for (int a = 0; a < A; ++a) {
#pragma unroll
        for (int e = 0; e < E; ++e) acc[a].v[e] = 0.f;
        wv[a] = ldf<BF>(P.w + (size_t)(a * P.no + P.dc) * P.Cin, qc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // Two register sets of kLaneRows rows each: the loads of batch k + 1 are requested BEFORE batch k is multiplied and stored, so a
    // workgroup that walks a tall tile (several batches) keeps its loads in flight across its own stores -- [r4] with that the tile
    // height is free, and the grid is cut into equal shares of the bytes (see pick_rows) instead of 2 368 short workgroups
    Vec<BF> xa[kLaneRows], xb[kLaneRows];
    auto load_rows = [&](Vec<BF> (&xv)[kLaneRows], int base) {
#pragma unroll
        for (int u = 0; u < kLaneRows; ++u) {
            const int r = min(rl + (base + u) * rows_par, nrows - 1);
            xv[u] = ldv<BF, true>(tx + ((unsigned)r * rowb + (unsigned)qc * 16u), 0);
        }
    };
    load_rows(xa, 0);
    if (tid < ng) s_g[tid] = g0 * f;
    for (int i = tid + kThreads; i < ng; i += kThreads) s_g[i] = P.gd[gd_index(P, n0 + i / A, i % A)] * f;
    lds_barrier();
    float* s_red = lds + P.rows_per_tile * A;
    if (cs == 0 && tid < A && P.part_b) {                // db partial of anchor `tid` over this row tile
        float s = 0.f;
        for (int r = 0; r < nrows; ++r) s += s_g[r * A + tid];
        P.part_b[(size_t)tile * A + tid] = s;
    }
    const int lane_rows = (P.rows_per_tile + rows_par - 1) / rows_par;
    auto consume = [&](const Vec<BF> (&xv)[kLaneRows], int base) {
#pragma unroll
        for (int u = 0; u < kLaneRows; ++u) {
            const int r = rl + (base + u) * rows_par;
            if (r < nrows && col_ok) {
                const float* g = s_g + r * A;
                Vec<BF> o;
#pragma unroll
                for (int e = 0; e < E; ++e) o.v[e] = 0.f;
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const float ga = g[a];
#pragma unroll
                    for (int e = 0; e < E; ++e) { o.v[e] = fmaf(ga, wv[a].v[e], o.v[e]); acc[a].v[e] = fmaf(ga, xv[u].v[e], acc[a].v[e]); }
                }
                if (want_x) stv<BF, false>(tg + ((unsigned)r * rowb + (unsigned)q * 16u), 0, o);   // plain stores: see the header
            }
        }
    };
    for (int base = 0; base < lane_rows; base += 2 * kLaneRows) {         // rows rl + k * rows_par, k ascending: the same order as ever
        if (base + kLaneRows < lane_rows) load_rows(xb, base + kLaneRows);
        consume(xa, base);
        if (base + kLaneRows >= lane_rows) break;
        if (base + 2 * kLaneRows < lane_rows) load_rows(xa, base + 2 * kLaneRows);
        consume(xb, base + kLaneRows);
    }
    if (want_w) {
        // combine the row lanes (fixed order) and write the tile's dW partial
#pragma unroll
        for (int a = 0; a < A; ++a)
#pragma unroll
            for (int e = 0; e < E; ++e) s_red[((rl * A + a) * E + e) * tpr + lane_c] = acc[a].v[e];
        lds_barrier();
        if (rl == 0 && col_ok) {
#pragma unroll
            for (int a = 0; a < A; ++a) {
                Vec<BF> s;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float t = s_red[(a * E + e) * tpr + lane_c];
                    for (int k = 1; k < rows_par; ++k) t += s_red[((k * A + a) * E + e) * tpr + lane_c];
                    s.v[e] = t;
                }
                stf<BF>(P.part_w + ((size_t)tile * A + a) * P.Cin, q, s);
            }
        }
    }
}

// ordered compaction of the rows t in [t_begin, t_end) that satisfy `pred(t, n, a)` into LDS lists (ascending t);
// returns the count (same in every thread).  t_end - t_begin <= cap.  s_wave: kThreads/64 ints of scratch.
template <typename Pred>
__device__ __forceinline__ int compact_rows(const Head& P, int cap, Pred pred, int* l_t, int* l_n, int* s_wave, int t_begin, int t_end) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int base = 0;
    for (int t0 = t_begin; t0 < t_end; t0 += kThreads) {
        const int t = t0 + tid;
        int n = -1, a = 0;
        if (t < t_end) cell_decode(P, P.row_cell[t], n, a);
        const bool keep = n >= 0 && pred(t, n, a);
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_wave[wv] = __popcll(m);
        __syncthreads();
        int off = base;
        for (int k = 0; k < wv; ++k) off += s_wave[k];
        int tot = 0;
        for (int k = 0; k < kThreads / 64; ++k) tot += s_wave[k];
        if (keep) {
            const int pos = off + __popcll(m & ((1ull << lane) - 1ull));
            if (pos < cap) { l_t[pos] = t; l_n[pos] = n; }
        }
        base += tot;
        __syncthreads();
    }
    return base < cap ? base : cap;
}

template <bool BF>
__global__ __launch_bounds__(kThreads) void headgrad_finish_kernel(const Multi M) {
    constexpr int E = Vec<BF>::N;
    extern __shared__ __align__(16) float lds[];
    __shared__ int s_wave[kThreads / 64];
    __shared__ float s_sum[kThreads / 64];
    const int tid = threadIdx.x, bid = blockIdx.x;
    int hi = 0;
    for (int i = 1; i < M.n; ++i) if (bid >= M.h[i].fd_blk0) hi = i;
    const Head& P = M.h[hi];
    const float f = P.upstream ? *P.upstream : 1.0f;
    if (bid < P.fd_blk0 + P.fd_nblk) {
        // ---- (a) dense channels: dW[a*no+dc, 4 float4] = sum over row tiles; 64 groups x 4 lanes, fixed order
        const int local = bid - P.fd_blk0;
        const int a = local / P.fd_vg, vg = local - a * P.fd_vg;
        const int ch = a * P.no + P.dc;
        const int l = tid & 3, grp = tid >> 2;
        const int q = vg * 4 + l;
        if (P.gw) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            if (q < P.cq) {
                int t = grp;
                for (; t + 192 < P.ntiles; t += 256) {                    // four partial tiles in flight
                    const float4 v0 = reinterpret_cast<const float4*>(P.part_w + ((size_t)t * P.A + a) * P.Cin)[q];
                    const float4 v1 = reinterpret_cast<const float4*>(P.part_w + ((size_t)(t + 64) * P.A + a) * P.Cin)[q];
                    const float4 v2 = reinterpret_cast<const float4*>(P.part_w + ((size_t)(t + 128) * P.A + a) * P.Cin)[q];
                    const float4 v3 = reinterpret_cast<const float4*>(P.part_w + ((size_t)(t + 192) * P.A + a) * P.Cin)[q];
                    s.x += v0.x; s.y += v0.y; s.z += v0.z; s.w += v0.w;
                    s.x += v1.x; s.y += v1.y; s.z += v1.z; s.w += v1.w;
                    s.x += v2.x; s.y += v2.y; s.z += v2.z; s.w += v2.w;
                    s.x += v3.x; s.y += v3.y; s.z += v3.z; s.w += v3.w;
                }
                for (; t < P.ntiles; t += 64) {
                    const float4 v = reinterpret_cast<const float4*>(P.part_w + ((size_t)t * P.A + a) * P.Cin)[q];
                    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                }
            }
            reinterpret_cast<float4*>(lds)[grp * 4 + l] = s;
            __syncthreads();
            if (grp < 8) {                                                 // 8 x 8 partial sums, then 8: still a fixed order
                float4 r = reinterpret_cast<float4*>(lds)[(grp * 8) * 4 + l];
                for (int k = 1; k < 8; ++k) { const float4 v = reinterpret_cast<float4*>(lds)[(grp * 8 + k) * 4 + l]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
                reinterpret_cast<float4*>(lds)[256 + grp * 4 + l] = r;
            }
            __syncthreads();
            if (grp == 0 && q < P.cq) {
                float4 r = reinterpret_cast<float4*>(lds)[256 + l];
                for (int k = 1; k < 8; ++k) { const float4 v = reinterpret_cast<float4*>(lds)[256 + k * 4 + l]; r.x += v.x; r.y += v.y; r.z += v.z; r.w += v.w; }
                reinterpret_cast<float4*>(P.gw + (size_t)ch * P.Cin)[q] = r;
            }
        }
        if (P.gb && vg == 0) {
            float s = 0.f;
            for (int t = tid; t < P.ntiles; t += kThreads) s += P.part_b[(size_t)t * P.A + a];
            s = wave_sum(s);
            __syncthreads();
            if ((tid & 63) == 0) s_sum[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) { float v = s_sum[0]; for (int k = 1; k < kThreads / 64; ++k) v += s_sum[k]; P.gb[ch] = v; }
        }
        return;
    }
    const int cap = M.cap;
    int* l_t = reinterpret_cast<int*>(lds);                 // [cap]
    int* l_n = l_t + cap;                                   // [cap]
    float* l_g = reinterpret_cast<float*>(l_n + cap);       // [cap] (channel blocks) / [no] (row blocks); cap >= no
    float* s_red = l_g + cap;                               // [kThreads * E]
    if (bid < P.fc_blk0 + P.fc_nblk) {
        // ---- (b) the other output channels: ch = a*no + c, c != dc; sum over the matched rows of anchor a.  The row list is
        //          walked in chunks of `cap` rows (the LDS lists): any number of rows, ascending row order within a thread group
        const int idx = bid - P.fc_blk0;
        const int a = idx / (P.no - 1);
        int c = idx - a * (P.no - 1); if (c >= P.dc) ++c;
        const int ch = a * P.no + c;
        int tpc = 1; while (tpc < P.cv && tpc < kThreads) tpc <<= 1;      // threads along Cin
        const int groups = kThreads / tpc, lane_c = tid & (tpc - 1), grp = tid / tpc;
        float bs = 0.f;
        for (int q0 = 0; q0 < P.cv; q0 += tpc) {
            const int q = q0 + lane_c;
            Vec<BF> s;
#pragma unroll
            for (int e = 0; e < E; ++e) s.v[e] = 0.f;
            for (int c0 = 0; c0 < P.n_rows; c0 += cap) {
                const int c1 = c0 + cap < P.n_rows ? c0 + cap : P.n_rows;
                const int cnt = compact_rows(P, cap, [&](int, int, int ta) { return ta == a; }, l_t, l_n, s_wave, c0, c1);
                for (int i = tid; i < cnt; i += kThreads) { const float g = P.gr[(size_t)l_t[i] * P.no + c] * f; l_g[i] = g; if (q0 == 0) bs += g; }
                __syncthreads();
                if (P.gw && q < P.cv) {
                    int i = grp;
                    for (; i + 3 * groups < cnt; i += 4 * groups) {       // four X rows in flight
                        Vec<BF> xv[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) xv[u] = ldv<BF>(P.x, (size_t)l_n[i + u * groups] * P.cv + q);
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const float g = l_g[i + u * groups];
#pragma unroll
                            for (int e = 0; e < E; ++e) s.v[e] += g * xv[u].v[e];
                        }
                    }
                    for (; i < cnt; i += groups) {
                        const Vec<BF> xv = ldv<BF>(P.x, (size_t)l_n[i] * P.cv + q);
                        const float g = l_g[i];
#pragma unroll
                        for (int e = 0; e < E; ++e) s.v[e] += g * xv.v[e];
                    }
                }
                __syncthreads();                                          // the next chunk overwrites the lists
            }
            if (!P.gw) break;
#pragma unroll
            for (int e = 0; e < E; ++e) s_red[(grp * E + e) * tpc + lane_c] = s.v[e];
            __syncthreads();
            if (grp == 0 && q < P.cv) {
                Vec<BF> r;
#pragma unroll
                for (int e = 0; e < E; ++e) {
                    float t = s_red[e * tpc + lane_c];
                    for (int k = 1; k < groups; ++k) t += s_red[(k * E + e) * tpc + lane_c];
                    r.v[e] = t;
                }
                stf<BF>(P.gw + (size_t)ch * P.Cin, q, r);
            }
            __syncthreads();
        }
        if (P.gb) {
            bs = wave_sum(bs);
            if ((tid & 63) == 0) s_sum[tid >> 6] = bs;
            __syncthreads();
            if (tid == 0) { float v = s_sum[0]; for (int k = 1; k < kThreads / 64; ++k) v += s_sum[k]; P.gb[ch] = v; }
        }
        return;
    }
    // ---- (c) dX of the matched cell-rows: one workgroup per row t0; the FIRST row of a cell-row n applies every row at n, in row
    //          order (several anchors of one cell, or YOLOv5's repeated rows, must not race).  The row is REWRITTEN, not
    //          corrected: dense part (the stream kernel's arithmetic, same order) + matched part summed in fp32, one store -- a
    //          bf16 dX is rounded once.
    if (!P.gx) return;
    const int t0 = bid - P.fr_blk0;
    if (t0 >= P.n_rows) return;
    int n0, a0; cell_decode(P, P.row_cell[t0], n0, a0);
    if (n0 < 0) return;                                                   // uniform
    int tpc = 1; while (tpc < P.cv && tpc < kThreads) tpc <<= 1;          // threads along Cin; the other thread groups split the channels
    const int groups = kThreads / tpc, lane_c = tid & (tpc - 1), grp = tid / tpc;
    for (int q0 = 0; q0 < P.cv; q0 += tpc) {
        const int q = q0 + lane_c;
        Vec<BF> s;
#pragma unroll
        for (int e = 0; e < E; ++e) s.v[e] = 0.f;
        bool first = true;
        for (int c0 = 0; c0 < P.n_rows; c0 += cap) {
            const int c1 = c0 + cap < P.n_rows ? c0 + cap : P.n_rows;
            const int cnt = compact_rows(P, cap, [&](int, int n, int) { return n == n0; }, l_t, l_n, s_wave, c0, c1);
            if (cnt == 0) continue;                                           // uniform (the lists are in LDS)
            if (first && l_t[0] != t0) return;                                // a lower row owns this cell-row
            first = false;
            for (int i = 0; i < cnt; ++i) {
                const int t = l_t[i];
                int n, a; cell_decode(P, P.row_cell[t], n, a);
                __syncthreads();
                for (int c = tid; c < P.no; c += kThreads) l_g[c] = c == P.dc ? 0.f : P.gr[(size_t)t * P.no + c] * f;
                __syncthreads();
                if (q < P.cv) {
                    const float* wb = P.w + (size_t)a * P.no * P.Cin;
                    int c = grp;
                    for (; c + 3 * groups < P.no; c += 4 * groups) {          // four weight rows in flight (slot dc carries 0)
                        const Vec<BF> w0 = ldf<BF>(wb + (size_t)c * P.Cin, q), w1 = ldf<BF>(wb + (size_t)(c + groups) * P.Cin, q),
                                      w2 = ldf<BF>(wb + (size_t)(c + 2 * groups) * P.Cin, q), w3 = ldf<BF>(wb + (size_t)(c + 3 * groups) * P.Cin, q);
                        const float g0 = l_g[c], g1 = l_g[c + groups], g2 = l_g[c + 2 * groups], g3 = l_g[c + 3 * groups];
#pragma unroll
                        for (int e = 0; e < E; ++e) s.v[e] += g0 * w0.v[e] + g1 * w1.v[e] + g2 * w2.v[e] + g3 * w3.v[e];
                    }
                    for (; c < P.no; c += groups) {
                        const Vec<BF> w0 = ldf<BF>(wb + (size_t)c * P.Cin, q);
                        const float g0 = l_g[c];
#pragma unroll
                        for (int e = 0; e < E; ++e) s.v[e] += g0 * w0.v[e];
                    }
                }
            }
            __syncthreads();                                                  // the next chunk overwrites the lists
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < E; ++e) s_red[(grp * E + e) * tpc + lane_c] = s.v[e];
        __syncthreads();
        if (grp == 0 && q < P.cv) {
            Vec<BF> r;                                                        // the dense part, as headgrad_stream_kernel computes it
#pragma unroll
            for (int e = 0; e < E; ++e) r.v[e] = 0.f;
            for (int a = 0; a < P.A; ++a) {
                const float ga = P.gd[gd_index(P, n0, a)] * f;
                const Vec<BF> wv = ldf<BF>(P.w + (size_t)(a * P.no + P.dc) * P.Cin, q);
#pragma unroll
                for (int e = 0; e < E; ++e) r.v[e] = fmaf(ga, wv.v[e], r.v[e]);
            }
#pragma unroll
            for (int e = 0; e < E; ++e) {
                float t = s_red[e * tpc + lane_c];
                for (int k = 1; k < groups; ++k) t += s_red[(k * E + e) * tpc + lane_c];
                r.v[e] += t;
            }
            stv<BF, false>(P.gx, (size_t)n0 * P.cv + q, r);
        }
        __syncthreads();
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct HPlan { int ntiles, rows_per_tile, tpr, rows_par, ncs, cv, cq; size_t part_w, part_b, bytes; size_t lds_stream; };

bool plan_one(const odpl_head_conv_t* h, const odpl_sparse_grad_t* g, int rows_per_tile, HPlan* p) {
    if (!h || !g || !h->x || !h->w) return false;
    const int E = h->bf16 ? 8 : 4;
    if (h->Cin < E || h->Cin % E != 0) return false;
    if (g->B < 1 || g->G2 < 1 || g->A < 1 || g->A > kMaxA || g->no < 2 || g->dense_ch < 0 || g->dense_ch >= g->no) return false;
    if (g->n_rows < 0 || g->no > kRowChunk) return false;          // any number of rows (chunked); the LDS lists also hold one row's `no` values
    if (g->layout != ODPL_LAYOUT_PLANAR && g->layout != ODPL_LAYOUT_INTERLEAVED) return false;
    if (!g->dense || (g->n_rows > 0 && (!g->rows || !g->row_cell))) return false;
    const long long N = (long long)g->B * g->G2;
    if (N * h->Cin > 0x7fffffffLL * 4LL || N > 0x7fffffffLL) return false;
    p->cv = h->Cin / E; p->cq = h->Cin / 4;
    int tpr = 1; while (tpr < p->cv && tpr < 64) tpr <<= 1;
    p->tpr = tpr; p->rows_par = kThreads / tpr;
    p->ncs = (p->cv + tpr - 1) / tpr;
    // `rows_per_tile` is quoted for 4 row lanes (>= 64 vectors per row); narrower rows have more row lanes per workgroup and get
    // proportionally taller tiles, so that every lane owns the same number of rows (8 at the default: one full set of loads)
    p->rows_per_tile = rows_per_tile * p->rows_par / 4;
    if (p->rows_per_tile < p->rows_par) p->rows_per_tile = p->rows_par;
    p->ntiles = (int)((N + p->rows_per_tile - 1) / p->rows_per_tile);
    if (g_rows_per_tile == 0) {                                  // equal shares: this head's N in ntiles tiles of (almost) the same height
        long long nt = (N + p->rows_per_tile / 2) / p->rows_per_tile; if (nt < 1) nt = 1;
        p->rows_per_tile = (int)((N + nt - 1) / nt);
        if (p->rows_per_tile < p->rows_par) p->rows_per_tile = p->rows_par;
        p->ntiles = (int)((N + p->rows_per_tile - 1) / p->rows_per_tile);
    }
    p->part_w = 0;
    p->part_b = align_up((size_t)p->ntiles * g->A * h->Cin * sizeof(float), 256);
    p->bytes = p->part_b + align_up((size_t)p->ntiles * g->A * sizeof(float), 256);
    p->lds_stream = ((size_t)p->rows_per_tile * g->A + (size_t)p->rows_par * g->A * tpr * E) * sizeof(float);
    return p->lds_stream <= 64 * 1024;
}

// rows per tile (quoted for 64-vector rows; narrower rows are scaled by plan_one) for a set of heads: the tunable, or [r4] the height
// at which the launch is g_share_wgs workgroups of equal bytes.  A workgroup streams rows x tpr vectors, so a head's cost is
// N * ncs * tpr vector-rows; the target per workgroup is the launch's total / g_share_wgs, and each head then cuts its own N into the
// number of tiles that comes closest (plan_one: heights differ by a row between heads, the bytes per workgroup by < 2 %).
// 2 368 workgroups of 32 rows left the last CUs with a tenth tile while others had nine, and wrote 7.2 MB of dW partials; 512 shares
// write 1.5 MB.
int pick_rows(int n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads) {
    if (g_rows_per_tile > 0) return g_rows_per_tile;
    double vec_rows = 0;
    for (int i = 0; i < n; ++i) {
        const int E = heads[i].bf16 ? 8 : 4, cv = heads[i].Cin / E;
        int tpr = 1; while (tpr < cv && tpr < 64) tpr <<= 1;
        vec_rows += (double)grads[i].B * grads[i].G2 * ((cv + tpr - 1) / tpr) * tpr;
    }
    int r = (int)(vec_rows / g_share_wgs / 64.0 + 0.5);     // rows of a 64-vector-wide workgroup
    if (r < 8) r = 8;
    if (r > 4096) r = 4096;
    return r;
}

template <bool BF>
int launch(const Multi& M, int A, int nstream, int nfinish, size_t lds_stream, hipStream_t stream) {
    prof_begin(ODPL_PROF_HEADGRAD_STREAM, stream);
    switch (A) {
        case 1: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 1>), dim3(nstream), dim3(kThreads), lds_stream, stream, M); break;
        case 2: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 2>), dim3(nstream), dim3(kThreads), lds_stream, stream, M); break;
        case 3: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 3>), dim3(nstream), dim3(kThreads), lds_stream, stream, M); break;
        case 4: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 4>), dim3(nstream), dim3(kThreads), lds_stream, stream, M); break;
        case 5: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 5>), dim3(nstream), dim3(kThreads), lds_stream, stream, M); break;
        default: return ODPL_EINVAL;
    }
    prof_end(ODPL_PROF_HEADGRAD_STREAM, stream);
    ODPL_HIP_TRY(hipGetLastError());
    size_t lds_finish = (size_t)M.cap * 12 + (size_t)kThreads * (BF ? 8 : 4) * sizeof(float);
    if (lds_finish < 288 * 16) lds_finish = 288 * 16;
    prof_begin(ODPL_PROF_HEADGRAD_FINISH, stream);
    hipLaunchKernelGGL((headgrad_finish_kernel<BF>), dim3(nfinish), dim3(kThreads), lds_finish, stream, M);
    prof_end(ODPL_PROF_HEADGRAD_FINISH, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" size_t odpl_head_conv_backward_multi_workspace_bytes(int32_t n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads) {
    if (n < 1 || n > ODPL_MAX_SCALES || !heads || !grads) return 0;
    size_t total = 0;
    for (int i = 0; i < n; ++i) if (!heads[i].x || heads[i].Cin < 4 || grads[i].B < 1 || grads[i].G2 < 1) return 0;
    const int rows = pick_rows(n, heads, grads);
    for (int i = 0; i < n; ++i) {
        HPlan p;
        if (!plan_one(&heads[i], &grads[i], rows, &p)) return 0;
        if (grads[i].A != grads[0].A || (heads[i].bf16 != 0) != (heads[0].bf16 != 0)) return 0;
        total += p.bytes;
    }
    return total ? total : 256;
}

extern "C" int odpl_head_conv_backward_multi(int32_t n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads,
                                             void* workspace, size_t workspace_bytes, void* stream_) {
    const size_t need = odpl_head_conv_backward_multi_workspace_bytes(n, heads, grads);
    if (need == 0 || !workspace || workspace_bytes < need || ((uintptr_t)workspace & 15) != 0) return ODPL_EINVAL;
    Multi M;
    M.n = n;
    // heaviest rows first: the workgroups of the widest head are the longest
    int order[ODPL_MAX_SCALES];
    for (int i = 0; i < n; ++i) order[i] = i;
    for (int i = 0; i < n; ++i) for (int j = i + 1; j < n; ++j) if (heads[order[j]].Cin > heads[order[i]].Cin) { const int t = order[i]; order[i] = order[j]; order[j] = t; }
    char* ws = (char*)workspace;
    size_t off = 0, lds_stream = 0;
    int nst = 0;
    const int rows = pick_rows(n, heads, grads);
    for (int k = 0; k < n; ++k) {
        const odpl_head_conv_t* h = &heads[order[k]];
        const odpl_sparse_grad_t* g = &grads[order[k]];
        HPlan pl;
        if (!plan_one(h, g, rows, &pl)) return ODPL_EINVAL;
        if (((uintptr_t)h->x & 15) || ((uintptr_t)h->w & 15) || ((uintptr_t)h->grad_x & 15) || ((uintptr_t)h->grad_w & 15)) return ODPL_EINVAL;
        Head& P = M.h[k];
        P.x = h->x; P.w = h->w; P.gx = h->grad_x; P.gw = h->grad_w; P.gb = h->grad_b;
        P.gd = g->dense; P.gr = g->rows; P.row_cell = g->row_cell; P.upstream = h->upstream;
        P.part_w = (float*)(ws + off + pl.part_w); P.part_b = (float*)(ws + off + pl.part_b);
        off += pl.bytes;
        P.N = g->B * g->G2; P.Cin = h->Cin; P.cv = pl.cv; P.cq = pl.cq; P.G2 = g->G2; P.A = g->A; P.no = g->no; P.dc = g->dense_ch;
        P.layout = g->layout; P.n_rows = g->n_rows;
        P.ntiles = pl.ntiles; P.rows_per_tile = pl.rows_per_tile; P.tpr = pl.tpr; P.rows_par = pl.rows_par; P.ncs = pl.ncs;
        P.st_blk0 = nst; P.st_nblk = pl.ntiles * pl.ncs; nst += P.st_nblk;
        if (pl.lds_stream > lds_stream) lds_stream = pl.lds_stream;
    }
    int nfin = 0, cap = 256;                 // LDS row-list capacity: >= every no, and every n_rows up to kRowChunk (longer lists: chunks)
    for (int k = 0; k < n; ++k) {
        Head& P = M.h[k];
        if (P.n_rows > cap) cap = (P.n_rows + 63) / 64 * 64;
        if (P.no > cap) cap = (P.no + 63) / 64 * 64;
        if (cap > kRowChunk) cap = kRowChunk;
        P.fd_vg = (P.cq + 3) / 4;
        P.fd_blk0 = nfin; P.fd_nblk = P.A * P.fd_vg; nfin += P.fd_nblk;
        P.fc_blk0 = nfin; P.fc_nblk = (P.gw || P.gb) ? P.A * (P.no - 1) : 0; nfin += P.fc_nblk;
        P.fr_blk0 = nfin; P.fr_nblk = P.gx ? P.n_rows : 0; nfin += P.fr_nblk;
    }
    M.cap = cap;
    hipStream_t stream = (hipStream_t)stream_;
    return heads[0].bf16 ? launch<true>(M, grads[0].A, nst, nfin, lds_stream, stream) : launch<false>(M, grads[0].A, nst, nfin, lds_stream, stream);
}

extern "C" size_t odpl_head_conv_backward_workspace_bytes(const odpl_head_conv_t* head, const odpl_sparse_grad_t* grad) {
    return odpl_head_conv_backward_multi_workspace_bytes(1, head, grad);
}

extern "C" int odpl_head_conv_backward(const odpl_head_conv_t* h, const odpl_sparse_grad_t* g, void* workspace,
                                       size_t workspace_bytes, void* stream) {
    return odpl_head_conv_backward_multi(1, h, g, workspace, workspace_bytes, stream);
}

extern "C" int odpl_head_conv_tune(int32_t rows_per_tile) {
    if (rows_per_tile != 0 && (rows_per_tile < 4 || rows_per_tile > 1024)) return ODPL_EINVAL;
    g_rows_per_tile = rows_per_tile;
    return 0;
}

extern "C" int odpl_head_conv_tune_shares(int32_t workgroups) {
    if (workgroups < 16 || workgroups > 65536) return ODPL_EINVAL;
    g_share_wgs = workgroups;
    return 0;
}
