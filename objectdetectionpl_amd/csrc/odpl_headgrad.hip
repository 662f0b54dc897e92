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
//   finish  : one launch, three kinds of workgroup, the longest dependent chains first in the grid:
//             (c) dX of the matched cell-rows, kTR = 8 matched rows x a slice of 32 / 64 16-byte columns per workgroup: the rows' gradient
//             values are laid out over the A*no output channels in LDS (several anchors of one cell, or YOLOv5's repeated rows,
//             add up there in row order) and dX[8 rows, slice] = G[8, A*no] . W[A*no, slice], the sum split over the workgroup's
//             thread groups and combined in a fixed order; a cell-row is REWRITTEN by the workgroup that holds its first row: the
//             dense part again (the stream kernel's arithmetic) + the matched part, summed in fp32 and stored once, so a bf16 dX
//             is rounded once.  W is read once per 8 rows (one workgroup per row read it 8x as often: 1.5 GB of L2 reads at config 5);
//             (b) the other output channels, kTC = 4 channels of one anchor x a slice of 32 columns per workgroup:
//             dW[4, slice] = g[4, rows of the anchor] . X[rows, slice], the rows split over the thread groups, 6 / 8 gathered X
//             rows in flight per lane;
//             (a) 64 x 4-lane groups per workgroup reduce the stream kernel's partials of the dense channels in a fixed order.
//             (b) and (c) find their rows by an ordered scan of the row list held in registers (one barrier; cell -> (row, anchor) by
//             multiply-and-shift, not division); lists longer than kRowChunk are walked in chunks: no limit on the number of rows.
//             A list sized for its capacity (YOLOv5: 5 * A * nT) carries its produced count in n_rows_dev: the scan stops there,
//             and tiles of dropped rows leave at once.  The launch is latency-bound (dependent gathers): 165 registers keep three
//             workgroups per CU, the LDS plan (<= 53 KB) is sized for the same three.  Config 5 (3 x ~1 000 rows of 3 360): 118 -> 57 us;
//             config 2 (3 x 112 rows): 20.5 -> 19 us.  tools/probe_headgrad_finish.py times the parts (-DODPL_FINISH_TRACE: per-phase stamps)
// Bitwise reproducible run to run.
#include "odpl_common.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

constexpr int kThreads = 256;
constexpr int kMaxA = 5;
constexpr int kRowChunk = 1792;         // matched rows the finish kernel stages in LDS at a time (longer row lists are walked in chunks)
constexpr int kTC = 4;                  // finish (b): output channels per workgroup (one float4 of gradient values per row)
constexpr int kTR = 8;                  // finish (c): matched rows per workgroup
constexpr int kKC = 512;                // finish (c): output channels per LDS slab of the rows' gradient matrix
constexpr int kFinRounds = 4;            // finish (c): rounds of the combine at most (kTR rows, >= 2 at a time)
constexpr int kStage = 4096;             // finish (c): floats of gradient rows staged in LDS at a time (>= the largest `no`)
constexpr int kFinCols = 32;            // finish (b), (c): 16-byte columns per workgroup (the other kThreads / kFinCols-fold split is over the sum)
int g_rows_per_tile = 32;              // rows of X per workgroup (measured best at config 2: 28.2 us; 0 = auto: one resident wave of workgroups, 29.8 us)

struct Head {
    const void* x; const float* w; void* gx; float* gw; float* gb;
    const float* gd; const float* gr; const int* row_cell; const int* n_dev; const float* upstream;
    float* part_w;        // [ntiles, A, Cin]
    float* part_b;        // [ntiles, A]
    int N, Cin, cv, cq, G2, A, no, dc, layout, n_rows;
    int ntiles, rows_per_tile, tpr, rows_par, ncs;
    int st_blk0, st_nblk;                 // stream kernel blocks
    int fd_blk0, fd_nblk, fd_vg;          // finish: dense-channel reduce blocks (fd_vg 4-float4 groups per anchor)
    int fc_blk0, fc_nblk, fc_tiles;       // finish: one block per kTC other output channels of an anchor (fc_tiles per anchor)
    int fr_blk0, fr_nblk;                 // finish: one block per kTR matched rows
    int fin_tpc, fin_ncs, fr_tpc, fr_ncs; // finish (b) / (c): lanes along Cin per workgroup, column slices per tile
    unsigned long long mG2, mA; int kG2, kA;     // cell / G2 and plane / A as a multiply and a shift (magic_of): the finish kernel decodes n_rows cells per workgroup
};
struct Multi { Head h[ODPL_MAX_SCALES]; int n, cap, gslab, stage, rd_off, fr_total, fc_total; };      // finish: LDS row-list capacity (rows per chunk), channels per slab of the gradient matrix, floats of staged gradient rows

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
// a 16-byte vector of X as loaded (kept raw while more loads are in flight) and its E floats
typedef unsigned u4raw __attribute__((ext_vector_type(4)));
template <bool BF>
__device__ __forceinline__ Vec<BF> unpack(const u4raw u) {
    Vec<BF> r;
    if (!BF) { r.v[0] = __uint_as_float(u[0]); r.v[1] = __uint_as_float(u[1]); r.v[2] = __uint_as_float(u[2]); r.v[3] = __uint_as_float(u[3]); }
    else {
        const unsigned w[4] = {u[0], u[1], u[2], u[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) { r.v[2 * k] = __uint_as_float(w[k] << 16); r.v[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u); }
    }
    return r;
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
// x / d for 0 <= x < 2^31 as (x * m) >> k with k = 32 + ceil(log2 d), m = floor(2^k / d) + 1: exact (x * (m*d - 2^k) <= x * d < 2^k)
// and x * m < 2^64
__device__ __forceinline__ int fast_div(int x, unsigned long long m, int k) { return (int)(((unsigned long long)(unsigned)x * m) >> k); }
// cell_decode without the integer divisions (same results)
__device__ __forceinline__ void cell_decode_fast(const Head& P, int cell, int& n, int& a) {
    if (cell < 0) { n = -1; a = 0; return; }
    if (P.layout == ODPL_LAYOUT_PLANAR) {
        const int plane = fast_div(cell, P.mG2, P.kG2), s = cell - plane * P.G2, b = fast_div(plane, P.mA, P.kA);
        a = plane - b * P.A; n = b * P.G2 + s;
    } else { n = fast_div(cell, P.mA, P.kA); a = cell - n * P.A; }
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
    Vec<BF> xv[kLaneRows];
    auto load_rows = [&](int base) {
#pragma unroll
        for (int u = 0; u < kLaneRows; ++u) {
            const int r = min(rl + (base + u) * rows_par, nrows - 1);
            xv[u] = ldv<BF, true>(tx + ((unsigned)r * rowb + (unsigned)qc * 16u), 0);
        }
    };
    load_rows(0);
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
    for (int base = 0;;) {
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
        base += kLaneRows;
        if (base >= lane_rows) break;
        load_rows(base);
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
// returns the count (same in every thread).  t_end - t_begin <= cap <= kRowChunk.  s_wave: kThreads/64 ints of scratch.
// Each wave takes a contiguous quarter of the range and keeps its rows' cells in registers (kRowChunk / kThreads loads per lane, all
// in flight at once: load_cells, which a caller may issue early); one barrier between counting and writing -- the scan is a few load
// latencies, not one per 256 rows.
struct Cells { int c[kRowChunk / kThreads]; };
// this lane's cells of the range [t_begin, t_end): issued here, first used by compact_cells
__device__ __forceinline__ Cells load_cells(const Head& P, int t_begin, int t_end) {
    constexpr int kPer = kRowChunk / kThreads;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int seg = (t_end - t_begin + kThreads - 1) / kThreads * 64;     // rows per wave, a multiple of 64
    const int w0 = t_begin + wv * seg;
    Cells r;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        const int t = w0 + u * 64 + lane;
        r.c[u] = (u * 64 < seg && t < t_end) ? P.row_cell[t] : -1;
    }
    return r;
}
template <typename Pred>
__device__ __forceinline__ int compact_cells(const Head& P, int cap, Pred pred, const Cells& cells, int* l_t, int* l_n, int* s_wave, int t_begin, int t_end,
                                            int* l_a = nullptr) {
    constexpr int kPer = kRowChunk / kThreads;                            // rows per lane
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int seg = (t_end - t_begin + kThreads - 1) / kThreads * 64;
    const int w0 = t_begin + wv * seg;
    const int* cell = cells.c;
    int nn[kPer], aa[kPer];
    unsigned long long keep[kPer];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        int n = -1, a = 0;
        keep[u] = 0ull;
        if (u * 64 < seg) {                                               // uniform
            cell_decode_fast(P, cell[u], n, a);
            keep[u] = __ballot(n >= 0 && pred(w0 + u * 64 + lane, n, a));
        }
        nn[u] = n; aa[u] = a;
        mine += __popcll(keep[u]);
    }
    if (lane == 0) s_wave[wv] = mine;
    __syncthreads();
    int off = 0, tot = 0;
    for (int k = 0; k < kThreads / 64; ++k) { const int c = s_wave[k]; if (k < wv) off += c; tot += c; }
#pragma unroll
    for (int u = 0; u < kPer; ++u) {
        if ((keep[u] >> lane) & 1ull) {
            const int pos = off + __popcll(keep[u] & ((1ull << lane) - 1ull));
            if (pos < cap) { l_t[pos] = w0 + u * 64 + lane; l_n[pos] = nn[u]; if (l_a) l_a[pos] = aa[u]; }
        }
        off += __popcll(keep[u]);
    }
    __syncthreads();
    return tot < cap ? tot : cap;
}

#ifdef ODPL_FINISH_TRACE
__device__ unsigned long long* g_trace = nullptr;                         // [nblocks][8] wall-clock stamps (100 MHz)
#define TRACE(slot) do { if (g_trace && threadIdx.x == 0) g_trace[(size_t)blockIdx.x * 8 + (slot)] = wall_clock64(); } while (0)
#else
#define TRACE(slot) do { } while (0)
#endif

template <bool BF>
__global__ __launch_bounds__(kThreads) void headgrad_finish_kernel(const Multi M) {
    constexpr int E = Vec<BF>::N;
    extern __shared__ __align__(16) float lds[];
    __shared__ int s_wave[kThreads / 64];
    __shared__ float s_sum[kThreads / 64 * kTC];
    const int tid = threadIdx.x, bid = blockIdx.x;
    TRACE(0);
    // grid order: every head's (c) blocks, then every head's (b) blocks, then the (a) blocks -- the longest dependent chains first
    const int part = bid < M.fr_total ? 2 : (bid < M.fr_total + M.fc_total ? 1 : 0);
    int hi = 0;
    for (int i = 1; i < M.n; ++i) if (bid >= (part == 2 ? M.h[i].fr_blk0 : part == 1 ? M.h[i].fc_blk0 : M.h[i].fd_blk0)) hi = i;
    const Head& P = M.h[hi];
    const float f = P.upstream ? *P.upstream : 1.0f;
    if (part == 0) {
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
        TRACE(7);
        return;
    }
    const int cap = M.cap;
    const Cells pre = load_cells(P, 0, P.n_rows < cap ? P.n_rows : cap);      // the first chunk of the row list: in flight from here
    int n_live = P.n_rows;                                                    // rows the chunk loops visit: the produced count when the list says
    if (P.n_dev) { const int nd = *P.n_dev; n_live = nd < n_live ? (nd > 0 ? nd : 0) : n_live; }
    int* l_t = reinterpret_cast<int*>(lds);                 // [cap]
    int* l_n = l_t + cap;                                   // [cap]
    // (b) and (c) are gathers with a dependent address per row: what bounds them is how many loads are in flight, so a workgroup
    // takes a slice of <= kFinCols 16-byte columns and splits its sum over kThreads / kFinCols thread groups.  The groups' sums meet
    // in LDS laid out as the output rows ([group][row][column]); every thread then adds one float4 column over the groups (fixed order).
    const bool is_b = part == 1;
    const int tpc = is_b ? P.fin_tpc : P.fr_tpc;                          // threads along Cin (power of two)
    const int groups = kThreads / tpc, lane_c = tid & (tpc - 1), grp = tid / tpc;
    const int cols = tpc * E, v4 = cols / 4;                              // floats / float4 per output row of this slice
    const int vs = tid / v4, vc = tid - vs * v4;                          // the combine: this thread's output row and float4 column
    if (is_b) {
        // ---- (b) the other output channels, kTC of one anchor per workgroup: dW[a*no + c, :] = sum over the matched rows of anchor a of
        //          g[t, c] * X[n_t, :] -- a [kTC x rows] x [rows x Cin] product with the rows gathered; every X row is read once per kTC
        //          channels.  The row list is walked in chunks of `cap` rows (the LDS lists): any number of rows, ascending row
        //          order within a thread group, the groups combined in a fixed order
        constexpr int kXU = BF ? 6 : 8;                                   // gathered X rows in flight per lane (with 8, the bf16 build loses a wave per SIMD)
        float4* l_g = reinterpret_cast<float4*>(l_n + cap);               // [cap] the rows' kTC gradient values
        float* s_red = lds;                                               // [groups][kTC][cols], over the lists once they are dead
        const int idx = (bid - P.fc_blk0) / P.fin_ncs, cs = (bid - P.fc_blk0) - idx * P.fin_ncs;
        const int a = idx / P.fc_tiles, ct = idx - a * P.fc_tiles;
        int cch[kTC]; bool cok[kTC];
#pragma unroll
        for (int u = 0; u < kTC; ++u) {
            const int k = ct * kTC + u;
            cok[u] = k < P.no - 1;
            cch[u] = cok[u] ? (k >= P.dc ? k + 1 : k) : 0;
        }
        float bs[kTC];
#pragma unroll
        for (int u = 0; u < kTC; ++u) bs[u] = 0.f;
        const int q = cs * tpc + lane_c;
        Vec<BF> s[kTC];
#pragma unroll
        for (int u = 0; u < kTC; ++u)
#pragma unroll
            for (int e = 0; e < E; ++e) s[u].v[e] = 0.f;
        for (int c0 = 0; c0 < n_live; c0 += cap) {
            const int c1 = c0 + cap < P.n_rows ? c0 + cap : P.n_rows;
            Cells cl = pre;
            if (c0 > 0) cl = load_cells(P, c0, c1);
            const int cnt = compact_cells(P, cap, [&](int, int, int ta) { return ta == a; }, cl, l_t, l_n, s_wave, c0, c1);
            TRACE(1);
            for (int i = tid; i < cnt; i += kThreads) {
                const float* gr = P.gr + (size_t)l_t[i] * P.no;
                float g[kTC];
#pragma unroll
                for (int u = 0; u < kTC; ++u) { g[u] = cok[u] ? gr[cch[u]] * f : 0.f; if (cs == 0) bs[u] += g[u]; }
                l_g[i] = make_float4(g[0], g[1], g[2], g[3]);
            }
            __syncthreads();
            TRACE(2);
            if (P.gw && q < P.cv) {
                int i = grp;
#pragma unroll 1
                for (; i + (kXU - 1) * groups < cnt; i += kXU * groups) {       // kXU X rows in flight
                    u4raw raw[kXU];
#pragma unroll
                    for (int u = 0; u < kXU; ++u) raw[u] = reinterpret_cast<const u4raw*>(P.x)[(size_t)l_n[i + u * groups] * P.cv + q];
#pragma unroll
                    for (int u = 0; u < kXU; ++u) {
                        const Vec<BF> xv = unpack<BF>(raw[u]);
                        const float4 g = l_g[i + u * groups];
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            s[0].v[e] += g.x * xv.v[e]; s[1].v[e] += g.y * xv.v[e];
                            s[2].v[e] += g.z * xv.v[e]; s[3].v[e] += g.w * xv.v[e];
                        }
                    }
                }
                for (; i < cnt; i += groups) {
                    const Vec<BF> xv = ldv<BF>(P.x, (size_t)l_n[i] * P.cv + q);
                    const float4 g = l_g[i];
#pragma unroll
                    for (int e = 0; e < E; ++e) {
                        s[0].v[e] += g.x * xv.v[e]; s[1].v[e] += g.y * xv.v[e];
                        s[2].v[e] += g.z * xv.v[e]; s[3].v[e] += g.w * xv.v[e];
                    }
                }
            }
            __syncthreads();                                          // the next chunk (or the combine) overwrites the lists
            TRACE(3);
        }
        if (P.gw) {
#pragma unroll
            for (int u = 0; u < kTC; ++u) stf<BF>(s_red + (size_t)(grp * kTC + u) * cols, lane_c, s[u]);
            __syncthreads();
            const int k = ct * kTC + vs;                                  // this thread's output channel (vs < kTC) and float4 column
            if (vs < kTC && k < P.no - 1 && cs * cols + vc * 4 < P.Cin) {
                const float4* src = reinterpret_cast<const float4*>(s_red) + (size_t)vs * v4 + vc;
                float4 r = src[0];
                for (int g = 1; g < groups; ++g) { const float4 t = src[(size_t)g * kTC * v4]; r.x += t.x; r.y += t.y; r.z += t.z; r.w += t.w; }
                const int c = k >= P.dc ? k + 1 : k;
                *reinterpret_cast<float4*>(P.gw + (size_t)(a * P.no + c) * P.Cin + cs * cols + vc * 4) = r;
            }
        }
        if (P.gb && cs == 0) {
#pragma unroll
            for (int u = 0; u < kTC; ++u) {
                const float v = wave_sum(bs[u]);
                if ((tid & 63) == 0) s_sum[(tid >> 6) * kTC + u] = v;
            }
            __syncthreads();
            if (tid < kTC && cok[tid]) {
                float v = s_sum[tid];
                for (int k = 1; k < kThreads / 64; ++k) v += s_sum[k * kTC + tid];
                P.gb[a * P.no + cch[tid]] = v;
            }
        }
        TRACE(7);
        return;
    }
    // ---- (c) dX of the matched cell-rows, kTR matched rows per workgroup: dX[n, :] = G[n, :] . W with G the rows' gradient values
    //          laid out over the A*no output channels (several anchors of one cell, or YOLOv5's repeated rows, add up in row order in
    //          G) -- a [kTR x A*no] x [A*no x Cin] product; W is read once per kTR rows.  A cell-row is written by the workgroup that
    //          holds its FIRST row.  The row is REWRITTEN, not corrected: dense part (the stream kernel's arithmetic, same order) +
    //          matched part summed in fp32, one store -- a bf16 dX is rounded once.
    if (!P.gx) return;
    __shared__ int s_n[kTR], s_slot[kTR], s_first[kTR], s_amask;
    constexpr int kWU = BF ? 2 : 8;                                       // weight rows in flight per lane (32 B each for bf16 activations): what the register budget of three waves per SIMD leaves
    constexpr int kHmax = 4;                                              // rows combined at a time (at most); kFinRounds rounds at most (v4 <= 128)
    const int cs = (bid - P.fr_blk0) % P.fr_ncs;
    const int t0 = (bid - P.fr_blk0) / P.fr_ncs * kTR;
    int* l_a = l_n + cap;                                                 // [cap] the rows' anchors
    float* G = reinterpret_cast<float*>(l_a + cap);                        // [M.gslab][kTR]
    float* stg = G + M.gslab * kTR;                                       // [M.stage] gradient rows on their way into G
    float* s_red = lds;                                                   // [groups][rh][cols], over the lists once they are dead
    if (tid < kTR) {
        int n = -1, a = 0;
        if (t0 + tid < P.n_rows) cell_decode(P, P.row_cell[t0 + tid], n, a);
        s_n[tid] = n; s_first[tid] = 0x7fffffff;
    }
    if (tid == 0) s_amask = 0;
    __syncthreads();
    {                                                                     // a tile of dropped rows (cell -1: the tail of a capacity-sized list)
        bool any = false;
#pragma unroll
        for (int k = 0; k < kTR; ++k) any |= s_n[k] >= 0;
        if (!any) return;                                                 // uniform
    }
    TRACE(5);
    if (tid < kTR) {                                                      // the slot of a cell-row: its first row in this tile
        int sl = -1;
        if (s_n[tid] >= 0) { sl = tid; for (int k = tid - 1; k >= 0; --k) if (s_n[k] == s_n[tid]) sl = k; }
        s_slot[tid] = sl;
    }
    // The thread groups' sums are combined kRH rows at a time, one float4 of one row per thread: rounds = kTR / rh, rh = rows whose
    // float4 columns fill the workgroup (<= kHmax).  The dense part of this thread's output vector of every round (row r*rh + vs, float4
    // column vc), as headgrad_stream_kernel computes it, is loaded and evaluated now: its latency is spent beside the row list's
    const int rh = kThreads / v4 < kHmax ? kThreads / v4 : kHmax;
    const int rounds = kTR / rh;
    const int col = cs * cols + vc * 4;
    float4* rd = reinterpret_cast<float4*>(lds + M.rd_off);              // [kFinRounds][kThreads]: kept in LDS (16 registers otherwise), not aliased
    float4 wd[kFinRounds][kMaxA]; float gdv[kFinRounds][kMaxA];          // in registers only until the first scan of the row list is done
#pragma unroll
    for (int r = 0; r < kFinRounds; ++r) {
        const int n0 = (r < rounds && vs < rh) ? s_n[r * rh + vs] : -1;
#pragma unroll
        for (int a = 0; a < kMaxA; ++a) {
            const bool ok = n0 >= 0 && a < P.A && col < P.Cin;
            gdv[r][a] = ok ? P.gd[gd_index(P, n0, a)] : 0.f;
            wd[r][a] = ok ? *reinterpret_cast<const float4*>(P.w + (size_t)(a * P.no + P.dc) * P.Cin + col) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    TRACE(6);
    const int q = cs * tpc + lane_c;
    auto scan = [&](const Cells& cl, int c0, int c1) {
        return compact_cells(P, cap, [&](int, int n, int) {
            bool hit = false;
#pragma unroll
            for (int k = 0; k < kTR; ++k) hit |= n == s_n[k];
            return hit; }, cl, l_t, l_n, s_wave, c0, c1, l_a);
    };
    int c0 = 0, c1 = cap < P.n_rows ? cap : P.n_rows;
    int cnt = scan(pre, c0, c1);                                          // the tile holds a live row: n_live > 0
#pragma unroll
    for (int r = 0; r < kFinRounds; ++r) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int a = 0; a < kMaxA; ++a) {
            if (a < P.A) {
                const float ga = gdv[r][a] * f;
                v.x = fmaf(ga, wd[r][a].x, v.x); v.y = fmaf(ga, wd[r][a].y, v.y); v.z = fmaf(ga, wd[r][a].z, v.z); v.w = fmaf(ga, wd[r][a].w, v.w);
            }
        }
        if (r < rounds) rd[r * kThreads + tid] = v;
    }
    Vec<BF> s[kTR];
#pragma unroll
    for (int i = 0; i < kTR; ++i)
#pragma unroll
        for (int e = 0; e < E; ++e) s[i].v[e] = 0.f;
    for (;; ) {
        if (c0 > 0) { const Cells cl = load_cells(P, c0, c1); cnt = scan(cl, c0, c1); }
        TRACE(1);
        if (cnt > 0) {                                                    // uniform (the lists are in LDS)
        for (int i = tid; i < cnt; i += kThreads) {                       // l_n: the row's slot and anchor from here on
            const int n = l_n[i], a = l_a[i];
            int sl = 0;
            for (int k = kTR - 1; k >= 0; --k) if (s_n[k] == n) sl = k;
            l_n[i] = sl | (a << 8);
            atomicOr(&s_amask, 1 << a);
            atomicMin(&s_first[sl], l_t[i]);                              // the cell-row's first row
        }
        __syncthreads();
        if (c1 >= t0 + kTR || c1 >= n_live) {                           // every row up to the tile's own has been seen: the owners are known
            bool any = false;
#pragma unroll
            for (int k = 0; k < kTR; ++k) any |= s_slot[k] == k && s_first[k] == t0 + k;
            if (!any) return;                                             // uniform: every cell-row of this tile belongs to a lower row
        }
        const int amask = s_amask;
        TRACE(2);
        for (int k0 = 0; k0 < P.A * P.no; k0 += M.gslab) {                // the output channels in slabs (one slab up to A*no = kKC)
            const int k1 = min(k0 + M.gslab, P.A * P.no);
            __syncthreads();
            for (int i = tid; i < (k1 - k0) * kTR; i += kThreads) G[i] = 0.f;
            const int mb = P.no <= M.stage ? M.stage / P.no : 1;          // rows staged at a time
            for (int j0 = 0; j0 < cnt; j0 += mb) {
                const int nb = min(mb, cnt - j0);
                __syncthreads();
                for (int e = tid; e < nb * P.no; e += kThreads) {         // the rows' gradient values: every load in flight at once
                    const int j = e / P.no, c = e - j * P.no;
                    stg[e] = P.gr[(size_t)l_t[j0 + j] * P.no + c] * f;
                }
                __syncthreads();
                for (int k = k0 + tid; k < k1; k += kThreads) {           // thread = output channel: its rows in ascending order
                    const int a = k / P.no, c = k - a * P.no;
                    if (c == P.dc) continue;
                    float* g = G + (k - k0) * kTR;
                    for (int j = 0; j < nb; ++j) {
                        const int sa = l_n[j0 + j];
                        if ((sa >> 8) == a) g[sa & 255] += stg[j * P.no + c];
                    }
                }
            }
            __syncthreads();
            TRACE(3);
            if (q < P.cv) {
#pragma unroll 1
                for (int a = 0; a < P.A; ++a) {
                    if (!((amask >> a) & 1)) continue;                    // no row of this anchor in the tile: its channels are zero
                    const int ka = a * P.no;
                    const int lo = max(k0, ka) - ka, hi_c = min(k1, ka + P.no) - ka;
                    const float* wb = P.w + (size_t)ka * P.Cin;
                    const float* gb = G + (ka - k0) * kTR;
                    int c = lo + grp;
#pragma unroll 1
                    for (; c + (kWU - 1) * groups < hi_c; c += kWU * groups) {     // kWU weight rows in flight (slot dc carries 0)
                        Vec<BF> wv[kWU];
#pragma unroll
                        for (int u = 0; u < kWU; ++u) wv[u] = ldf<BF>(wb + (size_t)(c + u * groups) * P.Cin, q);
#pragma unroll
                        for (int u = 0; u < kWU; ++u) {
                            const float4 g0 = reinterpret_cast<const float4*>(gb + (c + u * groups) * kTR)[0];
                            const float4 g1 = reinterpret_cast<const float4*>(gb + (c + u * groups) * kTR)[1];
                            const float gv[kTR] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                            for (int i = 0; i < kTR; ++i)
#pragma unroll
                                for (int e = 0; e < E; ++e) s[i].v[e] += gv[i] * wv[u].v[e];
                        }
                    }
#pragma unroll 1
                    for (; c < hi_c; c += groups) {
                        const Vec<BF> wv = ldf<BF>(wb + (size_t)c * P.Cin, q);
                        const float4 g0 = reinterpret_cast<const float4*>(gb + c * kTR)[0];
                        const float4 g1 = reinterpret_cast<const float4*>(gb + c * kTR)[1];
                        const float gv[kTR] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
#pragma unroll
                        for (int i = 0; i < kTR; ++i)
#pragma unroll
                            for (int e = 0; e < E; ++e) s[i].v[e] += gv[i] * wv.v[e];
                    }
                }
            }
        }
        }
        __syncthreads();                                                  // the next chunk overwrites the lists
        TRACE(4);
        c0 += cap;
        if (c0 >= n_live) break;
        c1 = c0 + cap < P.n_rows ? c0 + cap : P.n_rows;
    }
#pragma unroll 1
    for (int r = 0; r < rounds; ++r) {                                    // the thread groups' sums, rh rows at a time
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kTR; ++i)
            if (i >= r * rh && i < (r + 1) * rh) stf<BF>(s_red + (size_t)(grp * rh + (i - r * rh)) * cols, lane_c, s[i]);
        __syncthreads();
        const int i = r * rh + vs;
        if (vs < rh && col < P.Cin && s_slot[i] == i && s_first[i] == t0 + i) {       // a cell-row's first row: the row is written here
            float4 o = rd[r * kThreads + tid];
            const float4* src = reinterpret_cast<const float4*>(s_red) + (size_t)vs * v4 + vc;
            float4 m = src[0];
            for (int g = 1; g < groups; ++g) { const float4 t = src[(size_t)g * rh * v4]; m.x += t.x; m.y += t.y; m.z += t.z; m.w += t.w; }
            o.x += m.x; o.y += m.y; o.z += m.z; o.w += m.w;
            const size_t at = (size_t)s_n[i] * P.Cin + col;
            if (!BF) *reinterpret_cast<float4*>(reinterpret_cast<float*>(P.gx) + at) = o;
            else {
                uint2 pk;
                pk.x = bf16_rne(o.x) | (bf16_rne(o.y) << 16); pk.y = bf16_rne(o.z) | (bf16_rne(o.w) << 16);
                *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(P.gx) + at) = pk;
            }
        }
    }
    TRACE(7);
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
// multiplier and shift of fast_div for the divisor d >= 1
inline void magic_of(int d, unsigned long long* m, int* k) {
    int lg = 0; while ((1ll << lg) < d) ++lg;
    *k = 32 + lg;
    *m = (unsigned long long)(((unsigned __int128)1 << *k) / (unsigned)d) + 1ull;
}

struct HPlan { int ntiles, rows_per_tile, tpr, rows_par, ncs, cv, cq; size_t part_w, part_b, bytes; size_t lds_stream; };

bool plan_one(const odpl_head_conv_t* h, const odpl_sparse_grad_t* g, int rows_per_tile, HPlan* p) {
    if (!h || !g || !h->x || !h->w) return false;
    const int E = h->bf16 ? 8 : 4;
    if (h->Cin < E || h->Cin % E != 0) return false;
    if (g->B < 1 || g->G2 < 1 || g->A < 1 || g->A > kMaxA || g->no < 2 || g->dense_ch < 0 || g->dense_ch >= g->no) return false;
    if (g->n_rows < 0 || g->no > 4096) return false;               // any number of rows (the finish launch walks long lists in chunks)
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
    p->part_w = 0;
    p->part_b = align_up((size_t)p->ntiles * g->A * h->Cin * sizeof(float), 256);
    p->bytes = p->part_b + align_up((size_t)p->ntiles * g->A * sizeof(float), 256);
    p->lds_stream = ((size_t)p->rows_per_tile * g->A + (size_t)p->rows_par * g->A * tpr * E) * sizeof(float);
    return p->lds_stream <= 64 * 1024;
}

// rows per tile for a set of heads: the tunable, or (auto) the smallest height >= 16 for which all heads' workgroups are
// resident at once -- 256 CUs x 8 workgroups of 256 threads -- so that no CU is left with a ninth tile while others idle
int pick_rows(int n, const odpl_head_conv_t* heads, const odpl_sparse_grad_t* grads) {
    if (g_rows_per_tile > 0) return g_rows_per_tile;
    for (int r = 16; r < 1024; ++r) {
        long long blocks = 0;
        for (int i = 0; i < n; ++i) {
            const int E = heads[i].bf16 ? 8 : 4, cv = heads[i].Cin / E;
            int tpr = 1; while (tpr < cv && tpr < 64) tpr <<= 1;
            const long long N = (long long)grads[i].B * grads[i].G2;
            const long long rh = (long long)r * (kThreads / tpr) / 4;
            blocks += ((N + rh - 1) / rh) * ((cv + tpr - 1) / tpr);
        }
        if (blocks <= 2048) return r;
    }
    return 1024;
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
    // finish, LDS per workgroup: (a) 288 float4; (b) lists + a float4 per row, then the kTC-row combine over them; (c) lists + a slab of
    // the rows' gradient matrix + staged gradient rows, then the combine over them, and behind either the dense parts of its output.
    // Three workgroups per CU (the register budget) need <= 53 KB: kRowChunk is sized for that
    const size_t fb = (size_t)M.cap * 24, fc = (size_t)M.cap * 12 + ((size_t)M.gslab * kTR + M.stage) * sizeof(float);
    const size_t red = (size_t)kThreads * (BF ? 8 : 4) * 4 * sizeof(float);              // [groups][<= 4 rows][cols]
    const size_t rd = (size_t)kFinRounds * kThreads * sizeof(float4);
    size_t c_base = fc > red ? fc : red;
    c_base = (c_base + 15) / 16 * 16;
    size_t lds_finish = c_base + rd;
    if (fb > lds_finish) lds_finish = fb;
    if (red > lds_finish) lds_finish = red;
    if (lds_finish < 288 * 16) lds_finish = 288 * 16;
    Multi Mf = M;
    Mf.rd_off = (int)(c_base / sizeof(float));
    prof_begin(ODPL_PROF_HEADGRAD_FINISH, stream);
    hipLaunchKernelGGL((headgrad_finish_kernel<BF>), dim3(nfinish), dim3(kThreads), lds_finish, stream, Mf);
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
        P.gd = g->dense; P.gr = g->rows; P.row_cell = g->row_cell; P.n_dev = g->n_rows_dev; P.upstream = h->upstream;
        P.part_w = (float*)(ws + off + pl.part_w); P.part_b = (float*)(ws + off + pl.part_b);
        off += pl.bytes;
        P.N = g->B * g->G2; P.Cin = h->Cin; P.cv = pl.cv; P.cq = pl.cq; P.G2 = g->G2; P.A = g->A; P.no = g->no; P.dc = g->dense_ch;
        P.layout = g->layout; P.n_rows = g->n_rows;
        magic_of(P.G2, &P.mG2, &P.kG2); magic_of(P.A, &P.mA, &P.kA);
        P.ntiles = pl.ntiles; P.rows_per_tile = pl.rows_per_tile; P.tpr = pl.tpr; P.rows_par = pl.rows_par; P.ncs = pl.ncs;
        P.st_blk0 = nst; P.st_nblk = pl.ntiles * pl.ncs; nst += P.st_nblk;
        if (pl.lds_stream > lds_stream) lds_stream = pl.lds_stream;
    }
    int nfin = 0, cap = 256;                 // LDS row-list capacity: every n_rows up to kRowChunk (longer lists: chunks)
    for (int k = 0; k < n; ++k) {
        Head& P = M.h[k];
        if (P.n_rows > cap) cap = (P.n_rows + 63) / 64 * 64;
        if (cap > kRowChunk) cap = kRowChunk;
        P.fd_vg = (P.cq + 3) / 4;
        P.fc_tiles = (P.no - 1 + kTC - 1) / kTC;
        P.fin_tpc = 1; while (P.fin_tpc < P.cv && P.fin_tpc < kFinCols) P.fin_tpc <<= 1;
        P.fin_ncs = (P.cv + P.fin_tpc - 1) / P.fin_tpc;
    }
    // (c): columns per workgroup.  Narrow slices split a tile's product over more thread groups (a shorter chain of weight loads) but
    // repeat its scan of the row list in more workgroups: 32 columns while the (c) grid stays within one resident wave of workgroups
    // (config 2: 24.5 -> 19.0 us), 64 beyond (config 5: 58.6 us against 63.4 with 32).  tpc * E / 4 <= 128 float4 per output row
    int fr_cols = 32;
    {
        long long wg = 0;
        for (int k = 0; k < n; ++k) if (M.h[k].gx) wg += (long long)((M.h[k].n_rows + kTR - 1) / kTR) * ((M.h[k].cv + 31) / 32);
        if (wg > 512) fr_cols = 64;
    }
    for (int k = 0; k < n; ++k) {
        Head& P = M.h[k];
        P.fr_tpc = 1; while (P.fr_tpc < P.cv && P.fr_tpc < fr_cols) P.fr_tpc <<= 1;
        P.fr_ncs = (P.cv + P.fr_tpc - 1) / P.fr_tpc;
    }
    for (int k = 0; k < n; ++k) { Head& P = M.h[k]; P.fr_blk0 = nfin; P.fr_nblk = P.gx ? (P.n_rows + kTR - 1) / kTR * P.fr_ncs : 0; nfin += P.fr_nblk; }
    M.fr_total = nfin;
    for (int k = 0; k < n; ++k) { Head& P = M.h[k]; P.fc_blk0 = nfin; P.fc_nblk = (P.gw || P.gb) ? P.A * P.fc_tiles * P.fin_ncs : 0; nfin += P.fc_nblk; }
    M.fc_total = nfin - M.fr_total;
    for (int k = 0; k < n; ++k) { Head& P = M.h[k]; P.fd_blk0 = nfin; P.fd_nblk = P.A * P.fd_vg; nfin += P.fd_nblk; }
    M.cap = cap;
    M.gslab = 4; M.stage = 4;
    for (int k = 0; k < n; ++k) {
        const Head& P = M.h[k];
        const int ch = P.A * P.no < kKC ? P.A * P.no : kKC;
        if (ch > M.gslab) M.gslab = ch;
        int st = 16 * P.no < kStage / 2 ? 16 * P.no : kStage / 2;        // 16 rows at a time, within 8 KB ...
        if (st < P.no) st = P.no;                                         // ... but one row at least
        if (st > M.stage) M.stage = st;
    }
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

#ifdef ODPL_FINISH_TRACE
extern "C" int odpl_head_conv_trace(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_trace), &buf, sizeof(buf)); }
#endif

extern "C" int odpl_head_conv_tune(int32_t rows_per_tile) {
    if (rows_per_tile != 0 && (rows_per_tile < 4 || rows_per_tile > 1024)) return ODPL_EINVAL;
    g_rows_per_tile = rows_per_tile;
    return 0;
}
