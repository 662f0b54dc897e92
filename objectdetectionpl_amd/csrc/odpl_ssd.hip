// SSD loss for MI355X (gfx950): prior<->annotation matching (corner IoU, forced best-prior matches),
// offset encoding with variances 0.1/0.2, cross-entropy (or the reference's softmax-focal) per prior,
// 3:1 hard-negative mining and d loss/d (offsets, classes) -- SSDLoss, losses.py:159-315, accuracy.py:6-37.
//
// One workgroup (1024 threads) per image; everything about an image stays on one CU:
//   * lens[j]: number of rows of the j-th distinct image id (losses.py:286-288).  QUIRK kept: image j is
//     scored against the FIRST lens[j] rows of the whole target tensor (losses.py:299-300);
//   * per-prior best annotation (first max) + per-annotation best prior by block-wide (value desc, index asc)
//     reductions; forced matches applied in annotation order (last wins a shared prior, losses.py:215-216);
//   * hard negatives without a sort: the k-th largest loss among the negatives by bisection on the float
//     bits (losses are >= 0), ties taken in index order;
//   * fixed-order reductions, no float atomics.
// The tensors are small (8 732 x C per image): latency/occupancy bound, not HBM bound.  [r4] timed phase by phase with wall-clock
// stamps (tools/time_ssd_loss.py --trace, a -DODPL_SSD_TRACE build) at B = 32, 21 classes, 8 boxes per image: 375 -> 154 us --
//   image ids all-against-all by every thread out of LDS (40 -> 9 us); both matchings in one pass over the IoUs, eight annotations at a
//   time (60 -> 38); logits through an LDS tile with the next tile prefetched into registers, eight row elements in flight, the
//   hardware exponential (109 -> 57); the hard-negative threshold by a three-level radix select on the loss bits instead of 31 block-wide
//   counts (34 -> 18); loss sums and gradient rows over the compacted list of active priors, one thread per active prior (104 -> 20).
#include "odpl_common.hpp"
#include "odpl_boxes_dev.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 1024;
constexpr int kMaxT = 8192;          // target rows staged in LDS as flags

struct SParams {
    const float* off; const float* cls; float* g_off; float* g_cls;
    const float* priors; const float* target;
    int B, P, C, nT, cls_mode, coord_sl1, neg_ratio;
    float thresh;
    float* loss; int* ann; int* flag;         // [B*P] per prior: CE loss, annotation index, bit0 matched bit1 selected-negative
    int* best_prior;                          // [B*nT]
    int* list;                                // [B*P] scratch: the active priors of an image (dense mode; rows-only writes them to `active`)
    float* part;                              // [B*2]
    int* meta;                                // [0] new_batch_size, [1] status
    float* out3; int* out_counts;
    int tile_rows;                                      // rows of logits per LDS tile (0: C too wide)
    int rows_only; int* active; int* active_count;      // ODPL_SSD_ROWS_ONLY: gradients on the active rows only + their list per image
};

// Row maximum and sum of exponentials of C logits, eight loads in flight at a time (a plain `for k < C` loop waits for every load -- LDS
// or L1 -- before issuing the next: 42 round trips per 21-class row); same order of operations as the plain loops.
__device__ __forceinline__ void row_max_sumexp(const float* x, int C, float& mx, float& se) {
    mx = -INFINITY;
    for (int k0 = 0; k0 < C; k0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[k0 + u < C ? k0 + u : C - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < C) mx = fmaxf(mx, v[u]);
    }
    se = 0.f;
    for (int k0 = 0; k0 < C; k0 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = x[k0 + u < C ? k0 + u : C - 1];
#pragma unroll
        for (int u = 0; u < 8; ++u) if (k0 + u < C) se += __expf(v[u] - mx);
    }
}

__device__ __forceinline__ Box4 clamp_points(float cx, float cy, float w, float h) {
    Box4 b;                                   // losses.py:180-181
    b.x1 = fmaxf(cx - w / 2.0f, 0.0f); b.y1 = fmaxf(cy - h / 2.0f, 0.0f);
    b.x2 = fminf(cx + w / 2.0f, 1.0f); b.y2 = fminf(cy + h / 2.0f, 1.0f);
    return b;
}

// (value desc, index asc) with NaN treated as the largest value: torch.max's first-index rule
__device__ __forceinline__ bool better(float ov, int oi, float bv, int bi) {
    const bool onan = ov != ov, bnan = bv != bv;
    if (onan != bnan) return onan;
    if (!onan && ov != bv) return ov > bv;
    return oi < bi;
}

// block-wide integer sum (result broadcast)
__device__ __forceinline__ int block_sum_i(int v, int* sh) {
    v = wave_sum_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
    for (int w = 0; w < kT / 64; ++w) t += sh[w];
    return t;
}
__device__ __forceinline__ float block_sum_f(float v, float* sh) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < kT / 64; ++w) t += sh[w];
    return t;
}

#ifdef ODPL_SSD_TRACE
__device__ unsigned long long* g_ssd_trace = nullptr;                      // [B][16] wall-clock stamps (100 MHz)
#define STRACE(slot) do { if (g_ssd_trace && threadIdx.x == 0) g_ssd_trace[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define STRACE(slot) do { } while (0)
#endif

__global__ __launch_bounds__(kT) void ssd_image_kernel(const SParams P) {
    __shared__ unsigned char s_first[kMaxT];
    __shared__ int s_i[kT / 64];
    __shared__ float s_f[kT / 64];
    __shared__ float s_bv[kT / 64]; __shared__ int s_bi[kT / 64];
    __shared__ int s_run;
    STRACE(0);
    const int j = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int np = P.P, C = P.C, nT = P.nT;
    const size_t p0 = (size_t)j * np;
    float* loss = P.loss + p0; int* ann = P.ann + p0; int* flag = P.flag + p0;
    // ---- lens[j] (losses.py:286-288): distinct image ids in ascending order.  The ids are compared all against all: out of LDS (the
    //      logits' tile, free until the classification pass) instead of nT dependent trips to L2 per thread (40 us at 256 rows)
    extern __shared__ __align__(16) float s_tile[];
    int cD = 0, cM = 0;
    if (3 * nT <= P.tile_rows * C && nT <= kT) {
        // every thread takes part: thread (t, part) scans a slice of the rows for row t; LDS integer atomics combine the slices
        float* s_id = s_tile; int* s_dup = reinterpret_cast<int*>(s_tile) + nT; int* s_rank = s_dup + nT;
        for (int t = tid; t < nT; t += kT) { s_id[t] = P.target[(size_t)t * 6]; s_dup[t] = 0; s_rank[t] = 0; }
        __syncthreads();
        int tpad = 1; while (tpad < nT) tpad <<= 1;
        const int parts = kT / tpad, t = tid & (tpad - 1), part = tid / tpad;
        const int per = (nT + parts - 1) / parts, u0 = part * per, u1 = u0 + per < nT ? u0 + per : nT;
        if (t < nT) {
            const float x = s_id[t];
            int dup = 0;
            for (int u = u0; u < u1 && u < t; ++u) dup |= (s_id[u] == x);
            if (dup) atomicOr(&s_dup[t], 1);
        }
        __syncthreads();
        if (tid < nT) s_first[tid] = s_dup[tid] == 0;
        __syncthreads();
        if (t < nT) {
            const float x = s_id[t];
            int r = 0;
            for (int u = u0; u < u1; ++u) r += (s_first[u] && s_id[u] < x);
            if (r) atomicAdd(&s_rank[t], r);
        }
        __syncthreads();
        if (tid < nT) { cD = s_first[tid]; cM = (s_rank[tid] == j); }
    } else {
        for (int t = tid; t < nT; t += kT) {
            const float x = P.target[(size_t)t * 6];
            bool first = true;
            for (int u = 0; u < t; ++u) if (P.target[(size_t)u * 6] == x) { first = false; break; }
            s_first[t] = first;
        }
        __syncthreads();
        for (int t = tid; t < nT; t += kT) {
            const float x = P.target[(size_t)t * 6];
            cD += s_first[t];
            int rank = 0;
            for (int u = 0; u < nT; ++u) rank += (s_first[u] && P.target[(size_t)u * 6] < x);
            cM += (rank == j);
        }
    }
    const int D = block_sum_i(cD, s_i);
    const int M = block_sum_i(cM, s_i);
    const int new_bs = D < P.B ? D : P.B;                   // losses.py:296-298
    if (j == 0 && tid == 0) P.meta[0] = new_bs;
    const bool want = P.g_cls != nullptr;
    if (j >= D) {                                           // "continue": the image contributes nothing
        if (want && !P.rows_only) {
            for (size_t e = tid; e < (size_t)np * C; e += kT) P.g_cls[p0 * C + e] = 0.f;
            for (size_t e = tid; e < (size_t)np * 4; e += kT) P.g_off[p0 * 4 + e] = 0.f;
        }
        if (tid == 0) { P.part[j * 2] = 0.f; P.part[j * 2 + 1] = 0.f; if (want && P.rows_only) P.active_count[j] = 0; }
        return;
    }
    STRACE(1);
    int* bestp = P.best_prior + (size_t)j * (nT > 0 ? nT : 1);
    constexpr int kHoldP = 16, kAnn = 8;
    if (np <= kHoldP * kT) {
        // ---- both matchings in ONE pass over the IoUs (losses.py:210-214): eight annotations at a time, every IoU is computed once and
        //      feeds the prior's best annotation (first maximum over the annotations, held in registers across the chunks) and the
        //      annotation's best prior (first maximum over the priors: thread-local in ascending order, then (value desc, index asc)
        //      reductions).  Two barriers per eight annotations instead of two per annotation + a second sweep of the IoUs.
        __shared__ float s_cbv[kT / 64][kAnn]; __shared__ int s_cbi[kT / 64][kAnn];
        float pbest[kHoldP]; int pbi[kHoldP];
#pragma unroll
        for (int u = 0; u < kHoldP; ++u) { pbest[u] = 0.f; pbi[u] = 0; }
        for (int m0 = 0; m0 < M; m0 += kAnn) {
            Box4 ab[kAnn];
#pragma unroll
            for (int a = 0; a < kAnn; ++a) {
                const float* y = P.target + (size_t)(m0 + a < M ? m0 + a : M - 1) * 6;
                ab[a] = clamp_points(y[2], y[3], y[4], y[5]);
            }
            float bv[kAnn]; int bp[kAnn]; bool have[kAnn];
#pragma unroll
            for (int a = 0; a < kAnn; ++a) { bv[a] = -INFINITY; bp[a] = 0x7fffffff; have[a] = false; }
#pragma unroll
            for (int u = 0; u < kHoldP; ++u) {
                const int p = tid + u * kT;
                if (p < np) {
                    const float4 d = reinterpret_cast<const float4*>(P.priors)[p];
                    const Box4 pb = clamp_points(d.x, d.y, d.z, d.w);
#pragma unroll
                    for (int a = 0; a < kAnn; ++a) {
                        if (m0 + a < M) {
                            const float v = iou_corner(pb, ab[a]);
                            if (m0 + a == 0 || (pbest[u] == pbest[u] && (v > pbest[u] || v != v))) { pbest[u] = v; pbi[u] = m0 + a; }
                            if (!have[a] || (bv[a] == bv[a] && (v > bv[a] || v != v))) { bv[a] = v; bp[a] = p; have[a] = true; }
                        }
                    }
                }
            }
#pragma unroll
            for (int a = 0; a < kAnn; ++a) {
                for (int o = 32; o > 0; o >>= 1) {          // (value desc, index asc); NaN treated as largest
                    const float ov = __shfl_down(bv[a], o, 64); const int oi = __shfl_down(bp[a], o, 64);
                    if (better(ov, oi, bv[a], bp[a])) { bv[a] = ov; bp[a] = oi; }
                }
                if (lane == 0) { s_cbv[wv][a] = bv[a]; s_cbi[wv][a] = bp[a]; }
            }
            __syncthreads();
            if (tid < kAnn && m0 + tid < M) {
                float v = s_cbv[0][tid]; int i = s_cbi[0][tid];
                for (int w = 1; w < kT / 64; ++w) {
                    const float ov = s_cbv[w][tid]; const int oi = s_cbi[w][tid];
                    if (better(ov, oi, v, i)) { v = ov; i = oi; }
                }
                bestp[m0 + tid] = i;
            }
            __syncthreads();
        }
#pragma unroll
        for (int u = 0; u < kHoldP; ++u) {
            const int p = tid + u * kT;
            if (p < np) { ann[p] = pbi[u]; flag[p] = pbest[u] >= P.thresh ? 1 : 0; }      // losses.py:214
        }
        STRACE(2);
    } else {
    // ---- per prior: best annotation (max over dim 0, first index), losses.py:213
    for (int p = tid; p < np; p += kT) {
        const float4 d = reinterpret_cast<const float4*>(P.priors)[p];
        const Box4 pb = clamp_points(d.x, d.y, d.z, d.w);
        float best = 0.f; int bi = 0;
        for (int m = 0; m < M; ++m) {
            const float* y = P.target + (size_t)m * 6;
            const float v = iou_corner(pb, clamp_points(y[2], y[3], y[4], y[5]));
            if (m == 0 || (best == best && (v > best || v != v))) { best = v; bi = m; }
        }
        ann[p] = bi;
        flag[p] = best >= P.thresh ? 1 : 0;                 // losses.py:214
    }
    STRACE(2);
    // ---- per annotation: best prior (max over dim 1, first index), losses.py:210
    for (int m = 0; m < M; ++m) {
        const float* y = P.target + (size_t)m * 6;
        const Box4 ab = clamp_points(y[2], y[3], y[4], y[5]);
        float bv = -INFINITY; int bp = 0x7fffffff; bool have = false;
        for (int p = tid; p < np; p += kT) {
            const float4 d = reinterpret_cast<const float4*>(P.priors)[p];
            const float v = iou_corner(clamp_points(d.x, d.y, d.z, d.w), ab);
            if (!have || (bv == bv && (v > bv || v != v))) { bv = v; bp = p; have = true; }
        }
        for (int o = 32; o > 0; o >>= 1) {                  // (value desc, index asc); NaN treated as largest
            const float ov = __shfl_down(bv, o, 64); const int oi = __shfl_down(bp, o, 64);
            if (better(ov, oi, bv, bp)) { bv = ov; bp = oi; }
        }
        __syncthreads();
        if (lane == 0) { s_bv[wv] = bv; s_bi[wv] = bp; }
        __syncthreads();
        if (tid == 0) {
            float v = s_bv[0]; int i = s_bi[0];
            for (int w = 1; w < kT / 64; ++w) {
                const float ov = s_bv[w]; const int oi = s_bi[w];
                if (better(ov, oi, v, i)) { v = ov; i = oi; }
            }
            bestp[m] = i;
        }
    }
    }
    __syncthreads();
    if (tid == 0) {                                         // losses.py:215-216, sequential: last annotation wins
        for (int m = 0; m < M; ++m) { const int p = bestp[m]; flag[p] |= 1; ann[p] = m; }
    }
    __syncthreads();
    // ---- per prior classification loss (reduction='none'), N
    STRACE(3);
    // The logits go through an LDS tile: the rows are C floats (84 bytes at 21 classes), so a thread per row reading global memory
    // directly spreads every load instruction over 64 cache lines (160 us of the 400 this kernel took at B = 32); a tile is copied
    // flat (coalesced) and each thread then reads ITS row out of LDS (row stride C words: conflict-free for odd C).
    const int tile_rows = P.tile_rows;                      // 0: C too wide for a tile -> rows straight from global memory
    int cN = 0, bad = 0;
    // the NEXT tile's floats wait in registers while this one is worked on (a tile is <= 48 KB = 12 floats per thread): the trip to
    // memory of tile i + 1 runs under the arithmetic of tile i
    constexpr int kPre = 12;
    float pre[kPre];
    auto fetch = [&](int r0) {
        const int ne = (np - r0 < tile_rows ? np - r0 : tile_rows) * C;
        const float* src = P.cls + (p0 + r0) * C;
#pragma unroll
        for (int u = 0; u < kPre; ++u) { const int e = tid + u * kT; pre[u] = e < ne ? src[e] : 0.f; }
    };
    if (tile_rows) fetch(0);
    for (int r0 = 0; r0 < np; r0 += (tile_rows ? tile_rows : kT)) {
        int p;
        const float* x;
        int fl = 0, an = 0;                                 // this row's match: loaded before the barriers, its class under the arithmetic
        if (tile_rows) {
            const int nr = np - r0 < tile_rows ? np - r0 : tile_rows;
            if (tid < nr) { fl = flag[r0 + tid]; an = ann[r0 + tid]; }
            __syncthreads();                                // the previous tile is consumed
#pragma unroll
            for (int u = 0; u < kPre; ++u) { const int e = tid + u * kT; if (e < nr * C) s_tile[e] = pre[u]; }
            __syncthreads();
            if (r0 + tile_rows < np) fetch(r0 + tile_rows);
            if (tid >= nr) continue;
            p = r0 + tid; x = s_tile + (size_t)tid * C;
        } else {
            p = r0 + tid;
            if (p >= np) continue;
            x = P.cls + (p0 + p) * C;
            fl = flag[p]; an = ann[p];
        }
        const int matched = fl & 1;
        cN += matched;
        const float cls_id = matched ? P.target[(size_t)an * 6 + 1] : 0.f;
        float mx, se;
        row_max_sumexp(x, C, mx, se);
        long long yv = matched ? trunc_ll(cls_id) : 0;      // losses.py:252-253
        float l = 0.f;
        if (P.cls_mode == ODPL_CLS_CE) {
            if (yv < 0 || yv >= C) { bad = 1; yv = 0; }
            l = (mx + logf(se)) - x[yv];                   // -log_softmax(x)[y]
        } else {                                            // softmax focal, one-hot without the background column
            if (yv < 0 || yv > C) { bad = 1; yv = 0; }
            if (yv >= 1) {
                const float pc = fminf(fmaxf(expf(x[yv - 1] - mx) / se, 1e-7f), 1.0f - 1e-7f);
                const float om = 1.0f - pc;
                l = 0.25f * (-1.0f * logf(pc)) * (om * om);
            }
        }
        loss[p] = l;
    }
    const int N = block_sum_i(cN, s_i);
    bad = __syncthreads_or(bad);
    STRACE(4);
    // ---- hard negatives (CE only; in focal mode background rows are exactly 0)
    int k = 0; unsigned T = 0; int cgt = 0;
    if (P.cls_mode == ODPL_CLS_CE) {
        const int nneg = np - N;
        k = P.neg_ratio * N < nneg ? P.neg_ratio * N : nneg;
        if (k > 0) {
            unsigned lo = 0u, hi = 0x7f800000u;             // largest T with count(neg bits >= T) >= k
            // the negatives' loss bits of this thread's priors stay in registers over the ~31 steps (8 732 priors: nine per thread);
            // a matched prior is entered as 0, which no step counts (mid >= 1)
            constexpr int kHold = 16;
            const bool held = np <= kHold * kT;
            unsigned lb[kHold];
#pragma unroll
            for (int u = 0; u < kHold; ++u) {
                const int p = tid + u * kT;
                lb[u] = (held && p < np && !(flag[p] & 1)) ? __float_as_uint(loss[p]) : 0u;
            }
            if (held && P.tile_rows * C >= 2048) {
                // radix select on the loss bits (losses are >= 0: the bits order like the values): three levels of an LDS histogram of
                // 2 048 bins (the logits' tile is free now) + a suffix scan, instead of 31 bisection steps of a block-wide count each
                int* hist = reinterpret_cast<int*>(s_tile);
                __shared__ int s_selb[2];
#pragma unroll
                for (int u = 0; u < kHold; ++u) { const int p = tid + u * kT; if (!(held && p < np && !(flag[p] & 1))) lb[u] = 0xffffffffu; }   // not a negative: outside every range
                unsigned base = 0u; int span_log = 31, shift = 20, c_above = 0;
                for (;;) {
                    hist[tid] = 0; hist[tid + kT] = 0;
                    __syncthreads();
#pragma unroll
                    for (int u = 0; u < kHold; ++u) { const unsigned d = lb[u] - base; if (lb[u] >= base && (d >> span_log) == 0u) atomicAdd(&hist[d >> shift], 1); }
                    __syncthreads();
                    const int h0 = hist[2 * tid], h1 = hist[2 * tid + 1];
                    int inc = h0 + h1;                               // inclusive suffix over threads tid .. 1023
                    for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_down(inc, o, 64); if (lane + o < 64) inc += v; }
                    if (lane == 0) s_i[wv] = inc;
                    __syncthreads();
                    int above = 0;
                    for (int w2 = wv + 1; w2 < kT / 64; ++w2) above += s_i[w2];
                    const int S0 = c_above + above + inc, S1 = S0 - h0, S2 = S1 - h1;
                    if (S1 >= k && S2 < k) { s_selb[0] = 2 * tid + 1; s_selb[1] = S2; }
                    else if (S0 >= k && S1 < k) { s_selb[0] = 2 * tid; s_selb[1] = S1; }
                    __syncthreads();
                    base += (unsigned)s_selb[0] << shift; c_above = s_selb[1];
                    __syncthreads();
                    if (shift == 0) break;
                    span_log = shift; shift = shift > 11 ? shift - 11 : 0;
                }
                lo = base; cgt = c_above;                            // largest T with count(bits >= T) >= k; count(bits > T)
                if (lo > hi) lo = hi;                                // (a NaN loss: the bisection's upper end)
            } else {
                while (lo < hi) {
                    const unsigned mid = lo + (hi - lo + 1) / 2;
                    int c = 0;
                    if (held) {
#pragma unroll
                        for (int u = 0; u < kHold; ++u) c += lb[u] >= mid;
                    } else {
                        for (int p = tid; p < np; p += kT) c += (!(flag[p] & 1) && __float_as_uint(loss[p]) >= mid);
                    }
                    c = block_sum_i(c, s_i);
                    if (c >= k) lo = mid; else hi = mid - 1;
                }
                cgt = -1;
            }
            T = lo;
            if (cgt < 0 || T == 0x7f800000u) {
                int c = 0;
                for (int p = tid; p < np; p += kT) c += (!(flag[p] & 1) && __float_as_uint(loss[p]) > T);
                cgt = block_sum_i(c, s_i);
            }
            // select: all > T, then ties == T in index order until k
            if (tid == 0) s_run = cgt;
            __syncthreads();
            for (int base = 0; base < np; base += kT) {
                const int p = base + tid;
                const bool neg = p < np && !(flag[p] & 1);
                const unsigned u = neg ? __float_as_uint(loss[p]) : 0u;
                const bool tie = neg && u == T;
                const unsigned long long mk = __ballot(tie);
                if (lane == 0) s_i[wv] = __popcll(mk);
                __syncthreads();
                int offc = s_run, tot = 0;
                for (int w = 0; w < kT / 64; ++w) { if (w < wv) offc += s_i[w]; tot += s_i[w]; }
                if (neg && u > T) flag[p] |= 2;
                if (tie && offc + __popcll(mk & ((1ull << lane) - 1ull)) < k) flag[p] |= 2;
                __syncthreads();
                if (tid == 0) s_run += tot;
                __syncthreads();
            }
        }
    }
    STRACE(5);
    // ---- sums (fixed thread-strided order) and gradients
    const float w = 1.0f / ((float)N * (float)new_bs);
    float a_cls = 0.f, a_reg = 0.f;
    const bool rows_only = want && P.rows_only;
    if (want && !rows_only) {                               // the dense gradients' zeros as two flat (coalesced) streams; the active rows follow
        for (size_t e = tid; e < (size_t)np * C; e += kT) P.g_cls[p0 * C + e] = 0.f;
        for (size_t e = tid; e < (size_t)np * 4; e += kT) P.g_off[p0 * 4 + e] = 0.f;
        __syncthreads();                                    // (block-wide ordering of the stores to one address)
    }
    STRACE(6);
    // The active priors (positives + mined negatives: ~7 % of the priors) are listed first, in index order (ballot prefix over the block),
    // and the loss sums / gradient rows then run over the LIST, one thread per active prior: with a thread per prior every wave
    // walked the two softmax passes for its four or five active lanes (104 us of this kernel's 375 at B = 32).
    int* lst = (P.rows_only && P.active) ? P.active + p0 : P.list + p0;
    int n_act = 0;                                          // active priors listed so far (uniform)
    for (int pb = 0; pb < np; pb += kT) {
        const int p = pb + tid;
        const bool act = p < np && (flag[p] & 3);
        const unsigned long long mk = __ballot(act);
        __syncthreads();
        if (lane == 0) s_i[wv] = __popcll(mk);
        __syncthreads();
        int offc = n_act, tot = 0;
        for (int w2 = 0; w2 < kT / 64; ++w2) { if (w2 < wv) offc += s_i[w2]; tot += s_i[w2]; }
        if (act) lst[offc + __popcll(mk & ((1ull << lane) - 1ull))] = p;
        n_act += tot;
    }
    __syncthreads();                                        // the list is read back by other threads of the block
    for (int li = tid; li < n_act; li += kT) {
        const int p = lst[li];
        const int f = flag[p];
        const float* x = P.cls + (p0 + p) * C;
        float* gx = want ? P.g_cls + (p0 + p) * C : nullptr;
        float* go = want ? P.g_off + (p0 + p) * 4 : nullptr;
        if (f & 3) {
            a_cls += loss[p];
            if (want) {
                long long yv = (f & 1) ? trunc_ll(P.target[(size_t)ann[p] * 6 + 1]) : 0;
                float mx, se;
                row_max_sumexp(x, C, mx, se);
                if (P.cls_mode == ODPL_CLS_CE) {
                    if (yv < 0 || yv >= C) yv = 0;
                    for (int q0 = 0; q0 < C; q0 += 8) {
                        float v[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) v[u] = x[q0 + u < C ? q0 + u : C - 1];
#pragma unroll
                        for (int u = 0; u < 8; ++u) if (q0 + u < C) gx[q0 + u] = (__expf(v[u] - mx) / se - (q0 + u == yv ? 1.0f : 0.0f)) * w;
                    }
                } else {
                    if (yv < 0 || yv > C) yv = 0;
                    if (yv >= 1) {
                        const int lab = (int)yv - 1;
                        const float pc_raw = expf(x[lab] - mx) / se;
                        const float pc = fminf(fmaxf(pc_raw, 1e-7f), 1.0f - 1e-7f), om = 1.0f - pc, lg = logf(pc);
                        const bool pass = pc_raw >= 1e-7f && pc_raw <= 1.0f - 1e-7f;
                        const float g = pass ? 0.25f * (-(om * om) / pc + 2.0f * om * lg) : 0.0f;
                        for (int q = 0; q < C; ++q) gx[q] = g * pc_raw * ((q == lab ? 1.0f : 0.0f) - expf(x[q] - mx) / se) * w;
                    } else {
                        for (int q = 0; q < C; ++q) gx[q] = 0.f;
                    }
                }
            }
        }
        if (f & 1) {                                        // losses.py:220-236, 250
            const float4 d = reinterpret_cast<const float4*>(P.priors)[p];
            const float* y = P.target + (size_t)ann[p] * 6;
            float t[4];
            t[0] = (y[2] - d.x) / (d.z * 0.1f); t[1] = (y[3] - d.y) / (d.w * 0.1f);
            t[2] = logf(y[4] / d.z) / 0.2f;     t[3] = logf(y[5] / d.w) / 0.2f;
            for (int q = 0; q < 4; ++q) {
                const float dd = P.off[(p0 + p) * 4 + q] - t[q];
                a_reg += coord_elem(dd, P.coord_sl1);
                if (want) go[q] = coord_deriv(dd, P.coord_sl1) * w;
            }
        } else if (rows_only) {
            go[0] = go[1] = go[2] = go[3] = 0.f;            // a listed hard negative: its box row is read by the heads' backward
        }
    }
    STRACE(7);
    a_cls = block_sum_f(a_cls, s_f);
    a_reg = block_sum_f(a_reg, s_f);
    if (rows_only && tid == 0) P.active_count[j] = n_act;
    if (tid == 0) {
        P.part[j * 2] = a_cls / (float)N;                   // losses.py:272
        P.part[j * 2 + 1] = a_reg / (float)N;               // losses.py:273
        if (bad) atomicOr(&P.meta[1], 1);
    }
}

#ifdef ODPL_SSD_TRACE
extern "C" int odpl_ssd_trace(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_ssd_trace), &buf, sizeof(buf)); }
#endif

__global__ void ssd_finalize_kernel(const SParams P) {
    if (threadIdx.x != 0) return;
    float cl = 0.f, lo = 0.f;
    for (int j = 0; j < P.B; ++j) { cl += P.part[j * 2]; lo += P.part[j * 2 + 1]; }     // losses.py:306-307
    const float nb = (float)P.meta[0];
    lo = lo / nb; cl = cl / nb;                                                          // :312-313
    P.out3[0] = lo + cl; P.out3[1] = lo; P.out3[2] = cl;
    if (P.out_counts) { P.out_counts[0] = P.meta[0]; P.out_counts[1] = P.meta[1]; }
    P.meta[1] = 0;
}

// `active` cut by source level: one workgroup per level walks the images' lists in order (ballot prefix), keeps the priors of its
// range and rebases them to the level's own anchor numbering
struct LevelP { const int* active; const int* active_count; int* lists; int* counts; int B, P, nlevels; int first[9]; };
__global__ __launch_bounds__(256) void ssd_level_lists_kernel(const LevelP L) {
    __shared__ int s_w[4];
    const int l = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int lo = L.first[l], hi = L.first[l + 1], an = hi - lo;
    int* out = L.lists + (size_t)L.B * lo;
    int n = 0;
    for (int b = 0; b < L.B; ++b) {
        const int cnt = L.active_count[b];
        const int* seg = L.active + (size_t)b * L.P;
        for (int i0 = 0; i0 < cnt; i0 += 256) {
            const int i = i0 + tid;
            const int p = i < cnt ? seg[i] : -1;
            const bool keep = p >= lo && p < hi;
            const unsigned long long mk = __ballot(keep);
            __syncthreads();
            if (lane == 0) s_w[wv] = __popcll(mk);
            __syncthreads();
            int off = n, tot = 0;
            for (int w = 0; w < 4; ++w) { if (w < wv) off += s_w[w]; tot += s_w[w]; }
            if (keep) out[off + __popcll(mk & ((1ull << lane) - 1ull))] = b * an + (p - lo);
            n += tot;
        }
    }
    if (tid == 0) L.counts[l] = n;
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
struct SPlan { size_t meta, loss, ann, flag, list, best, part, total; };
bool make_splan(const odpl_ssd_loss_desc_t* d, SPlan* p) {
    if (!d || d->B < 1 || d->P < 1 || d->C < 1 || d->nT < 1 || d->nT > kMaxT) return false;
    if (!d->off || !d->cls || !d->priors || !d->target) return false;
    if ((d->grad_off == nullptr) != (d->grad_cls == nullptr)) return false;
    if (d->flags & ~ODPL_SSD_ROWS_ONLY) return false;
    if ((d->flags & ODPL_SSD_ROWS_ONLY) && (!d->grad_off || !d->active || !d->active_count)) return false;
    const size_t BP = (size_t)d->B * d->P;
    size_t off = 0;
    p->meta = off; off += 256;
    p->loss = off; off += align_up(BP * 4, 256);
    p->ann = off;  off += align_up(BP * 4, 256);
    p->flag = off; off += align_up(BP * 4, 256);
    p->list = off; off += align_up(BP * 4, 256);
    p->best = off; off += align_up((size_t)d->B * d->nT * 4, 256);
    p->part = off; off += align_up((size_t)d->B * 8, 256);
    p->total = off;
    return true;
}
}  // namespace

extern "C" size_t odpl_ssd_loss_workspace_bytes(const odpl_ssd_loss_desc_t* d) {
    SPlan p;
    return make_splan(d, &p) ? p.total : 0;
}

extern "C" int odpl_ssd_loss(const odpl_ssd_loss_desc_t* d, void* workspace, size_t workspace_bytes,
                             float* out3, int32_t* out_counts, void* stream_) {
    SPlan pl;
    if (!make_splan(d, &pl) || !workspace || workspace_bytes < pl.total || !out3) return ODPL_EINVAL;
    if (((uintptr_t)workspace | (uintptr_t)d->priors) & 15) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    SParams P;
    P.off = d->off; P.cls = d->cls; P.g_off = d->grad_off; P.g_cls = d->grad_cls; P.priors = d->priors; P.target = d->target;
    P.B = d->B; P.P = d->P; P.C = d->C; P.nT = d->nT; P.cls_mode = d->cls_mode; P.coord_sl1 = d->coord_mode == ODPL_COORD_SMOOTH_L1;
    P.neg_ratio = 3; P.thresh = 0.5f;
    P.loss = (float*)(ws + pl.loss); P.ann = (int*)(ws + pl.ann); P.flag = (int*)(ws + pl.flag); P.list = (int*)(ws + pl.list);
    P.best_prior = (int*)(ws + pl.best); P.part = (float*)(ws + pl.part); P.meta = (int*)(ws + pl.meta);
    P.out3 = out3; P.out_counts = out_counts;
    P.rows_only = (d->flags & ODPL_SSD_ROWS_ONLY) ? 1 : 0; P.active = d->active; P.active_count = d->active_count;
    ODPL_HIP_TRY(odpl::fill_bytes(P.meta, 0, 256, stream));
    {   // LDS tile of the logits: <= 48 KB, a multiple of 64 rows, at most one row per thread
        int rows = (48 * 1024) / (4 * d->C);
        rows = rows / 64 * 64;
        if (rows > kT) rows = kT;
        P.tile_rows = rows >= 64 ? rows : 0;
    }
    hipLaunchKernelGGL(ssd_image_kernel, dim3(d->B), dim3(kT), (size_t)P.tile_rows * d->C * sizeof(float), stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ssd_finalize_kernel, dim3(1), dim3(64), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_ssd_level_lists(const int32_t* active, const int32_t* active_count, int32_t B, int32_t P, int32_t nlevels, const int32_t* first,
                                    int32_t* lists, int32_t* counts, void* stream) {
    if (!active || !active_count || !first || !lists || !counts || B < 1 || P < 1 || nlevels < 1 || nlevels > 8) return ODPL_EINVAL;
    if ((long long)B * P > 0x7fffffffLL) return ODPL_EINVAL;
    LevelP L;
    L.active = active; L.active_count = active_count; L.lists = lists; L.counts = counts; L.B = B; L.P = P; L.nlevels = nlevels;
    for (int l = 0; l <= nlevels; ++l) {
        L.first[l] = first[l];
        if (first[l] < 0 || first[l] > P || (l > 0 && first[l] <= first[l - 1])) return ODPL_EINVAL;
    }
    if (first[0] != 0 || first[nlevels] != P) return ODPL_EINVAL;
    hipLaunchKernelGGL(ssd_level_lists_kernel, dim3(nlevels), dim3(256), 0, (hipStream_t)stream, L);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
