// Batched NMS for MI355X (gfx950).
//
// (1) YOLO confidence-weighted MERGE NMS -- model/YOLOV4.py:221-283 (== YOLOV3.py:273-335,
//     YOLOV2.py:159-222, and YOLOV5.py:157-218 on its reinterpreted memory).  The reference flattens the
//     RAW heads (no sigmoid, no grid decode), thresholds conf at -0.0151, sorts by conf*max(cls) and loops
//     {IoU(+1) > thr & same label -> conf-weighted merge into the pivot, drop}.
//       prep   (N/64 x B WGs)   reads the NCHW heads once (coalesced along the G*G plane; each wave issues its whole class
//                               range before the first compare), class max/argmax per candidate, writes box/conf/cls_conf/
//                               label/score as SoA                                                     [HBM read bound]
//       image  (1 WG / image)   counts + compacts the valid candidates into LDS, LDS bitonic sort of (score desc, index asc),
//                               gather into rank order, class histogram/scan, then ONE WAVE PER CLASS runs that class's pivot
//                               loop (the same-label condition makes the classes independent) with its list in registers;
//                               rows land at the pivot's rank and a block scan compacts them in rank order.
//     The pivot is always retired (the reference never terminates when a box's self-IoU <= thr).
// (2) RetinaNet / SSD greedy top-k NMS -- model/RetinaNet.py:117-178 == SSD.py:249-310.
//       prep   (An/64 x B WGs)  class rows staged through LDS (coalesced 16-B loads, padded rows), sigmoid-max
//                               + first argmax per anchor, per-tile count of score > class_thresh
//       image  (1 WG / image)   block scan of the tile counts, barrier-free stable compaction, k-th largest by bisection on
//                               the float bits with the keys in registers (top-k without sorting the candidates), LDS
//                               bitonic sort of the <= topk survivors, greedy suppression by one wave, the reference's output
//                               quirks (keep indexes the FILTERED list, boxes/labels come from the UNFILTERED arrays; the
//                               last single box is dropped).
#include <limits.h>
#include "odpl_common.hpp"
#include "odpl_boxes_dev.hpp"
#include "odpl_profile.hpp"
#include "odpl_sort_dev.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

constexpr int kPrepT = 256;
constexpr int kPrepU = 24;             // class planes loaded per lane before the first compare
constexpr int kImgT = 1024;

struct YParams {
    const float* head[ODPL_MAX_SCALES];
    int G[ODPL_MAX_SCALES], first[ODPL_MAX_SCALES + 1];
    int nscales, B, A, no, N, Npad;
    float conf_thres, nms_thres;
    // per image SoA, unsorted then sorted
    float4* box; float* conf; float* cconf; int* label;
    float4* sbox; float* sconf; float* scconf; int* slabel;
    float* key; unsigned* idx;           // [B*Npad] sort buffers
    float* out; int* out_counts;
    // [r4] class-split pipeline (bucket -> class -> order)
    int* cls_off; int* cls_cnt;          // [B*kMaxCls] first slot / number of valid candidates of each class
    int* nvalid;                         // [B]
    int* ktot;                           // [B*kKtotPitch] rows kept so far per image: every class reserves a run of the compact kept list
    int* ksrc;                           // [B*Npad] compact kept list: the row's slot in krow (kkey / kidx: its pivot's score / candidate)
    float* ckey; unsigned* cidx;         // [B*Npad] (score, candidate) pairs grouped by class
    float* kkey; unsigned* kidx;         // [B*Npad] compact kept list: (score, candidate) of every kept pivot, class runs in arrival order
    float* krow;                         // [B*N*7] merged rows: class c's rows at the front of the class's own slots cls_off[c] ..
};

// A workgroup = 256 / SPLIT consecutive candidates; thread group q scans classes [q*nc/SPLIT, (q+1)*nc/SPLIT).  Lanes run
// along the G*G plane => every wave load is one coalesced 256-B segment.  Group 0 merges the partial (max, first argmax) pairs
// through LDS and writes the SoA.
template <int SPLIT>
__global__ __launch_bounds__(kPrepT) void yolo_nms_prep_kernel(const YParams P) {
    constexpr int CELLS = kPrepT / SPLIT;
    __shared__ float s_best[SPLIT][CELLS];
    __shared__ int s_bi[SPLIT][CELLS];
    const int b = blockIdx.y, cell = threadIdx.x % CELLS, q = threadIdx.x / CELLS, n = blockIdx.x * CELLS + cell;
    const int nc = P.no - 5;
    const float* base = nullptr;
    int G2 = 1;
    if (n < P.N) {
        int s = 0;
        while (s + 1 < P.nscales && n >= P.first[s + 1]) ++s;
        G2 = P.G[s] * P.G[s];
        const int r = n - P.first[s], a = r / G2, sp = r - a * G2;
        base = P.head[s] + ((size_t)(b * P.A + a) * P.no) * G2 + sp;
    }
    const int k0 = (int)(((long long)nc * q) / SPLIT), k1 = (int)(((long long)nc * (q + 1)) / SPLIT);
    float best = -INFINITY; int bi = 0x7fffffff; bool have = false;
    float bx_[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (base) {
        // few trips to memory: up to kPrepU class planes per lane (the whole range at SPLIT = 4, C = 80) and, in group 0, the
        // five box / confidence planes are issued before anything is consumed
        if (q == 0) {
#pragma unroll
            for (int u = 0; u < 5; ++u) bx_[u] = base[(size_t)u * G2];
        }
        for (int k = k0; k < k1; k += kPrepU) {
            float v[kPrepU];
#pragma unroll
            for (int u = 0; u < kPrepU; ++u) { const int kk = k + u < k1 ? k + u : k1 - 1; v[u] = base[(size_t)(5 + kk) * G2]; }
#pragma unroll
            for (int u = 0; u < kPrepU; ++u)
                if (k + u < k1 && (!have || (best == best && (v[u] > best || v[u] != v[u])))) { best = v[u]; bi = k + u; have = true; }
        }
    }
    if (SPLIT > 1) {
        s_best[q][cell] = best; s_bi[q][cell] = have ? bi : 0x7fffffff;
        __syncthreads();
    }
    if (q == 0) {
        if (base) {
            float bv = best; int bk = have ? bi : 0x7fffffff;
            for (int w = 1; w < SPLIT; ++w) {                                // class ranges ascend with w: first index wins ties
                const float ov = s_best[w][cell]; const int ok = s_bi[w][cell];
                if (ok == 0x7fffffff) continue;
                if (bk == 0x7fffffff || (bv == bv && (ov > bv || ov != ov))) { bv = ov; bk = ok; }
            }
            const float x = bx_[0], y = bx_[1], w = bx_[2], h = bx_[3];
            const float cf = bx_[4];
            const Box4 bx = corners_of(x, y, w, h);                          // xywh2xyxy, YOLOV4.py:253
            const size_t o = (size_t)b * P.N + n;
            P.box[o] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
            P.conf[o] = cf; P.cconf[o] = bv; P.label[o] = bk;
            const int valid = cf >= P.conf_thres;                            // YOLOV4.py:258
            P.key[(size_t)b * P.Npad + n] = valid ? cf * bv : -INFINITY;    // YOLOV4.py:263
            P.idx[(size_t)b * P.Npad + n] = valid ? (unsigned)n : 0xffffffffu;
        } else if (n < P.Npad) {
            P.key[(size_t)b * P.Npad + n] = -INFINITY;
            P.idx[(size_t)b * P.Npad + n] = 0xffffffffu;
        }
    }
}

// Fallback (nv > kCap or more than kMaxCls classes): global-memory sort + one block-wide sweep per pivot.
__device__ void yolo_nms_image_serial(const YParams& P, int nv) {
    __shared__ float s_red[6 * (kImgT / 64)];
    __shared__ int s_min[kImgT / 64];
    __shared__ int s_p;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* key = P.key + (size_t)b * P.Npad;
    unsigned* idx = P.idx + (size_t)b * P.Npad;
    float* out = P.out + (size_t)b * P.N * 7;
    if (nv == 0) {                                             // "continue": output[image_i] stays None
        if (tid == 0) P.out_counts[b] = -1;
        return;
    }
    bitonic_sort_global(key, idx, P.Npad);
    const size_t o0 = (size_t)b * P.N;
    for (int r = tid; r < nv; r += kImgT) {                    // gather into sorted order
        const unsigned n = idx[r];
        P.sbox[o0 + r] = P.box[o0 + n]; P.sconf[o0 + r] = P.conf[o0 + n];
        P.scconf[o0 + r] = P.cconf[o0 + n]; P.slabel[o0 + r] = P.label[o0 + n];
    }
    __syncthreads();
    const float4* sbox = P.sbox + o0; const float* sconf = P.sconf + o0; int* slabel = P.slabel + o0;
    int p = 0, K = 0;
    while (p < nv) {
        const float4 tb = sbox[p];
        const int lt = slabel[p];
        Box4 top; top.x1 = tb.x; top.y1 = tb.y; top.x2 = tb.z; top.y2 = tb.w;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int nxt = INT_MAX;
        for (int j = p + tid; j < nv; j += kImgT) {
            const int l = slabel[j];
            if (l < 0) continue;
            const float4 q = sbox[j];
            Box4 bj; bj.x1 = q.x; bj.y1 = q.y; bj.x2 = q.z; bj.y2 = q.w;
            const bool inv = (l == lt) && (iou_plus1(top, bj) > P.nms_thres);     // YOLOV4.py:271-274
            if (inv) {
                const float w = sconf[j];                                           // YOLOV4.py:275-277
                acc[0] += w; acc[1] += w * q.x; acc[2] += w * q.y; acc[3] += w * q.z; acc[4] += w * q.w;
                slabel[j] = -1;
            } else if (j != p) {
                nxt = min(nxt, j);
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = wave_sum(acc[i]);
        for (int o = 32; o > 0; o >>= 1) nxt = min(nxt, __shfl_down(nxt, o, 64));
        if (lane == 0) { for (int i = 0; i < 5; ++i) s_red[wv * 6 + i] = acc[i]; s_min[wv] = nxt; }
        __syncthreads();
        if (tid == 0) {
            float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f}; int mn = INT_MAX;
            for (int w = 0; w < kImgT / 64; ++w) { for (int i = 0; i < 5; ++i) a[i] += s_red[w * 6 + i]; mn = min(mn, s_min[w]); }
            float* row = out + (size_t)K * 7;
            row[0] = a[1] / a[0]; row[1] = a[2] / a[0]; row[2] = a[3] / a[0]; row[3] = a[4] / a[0];
            row[4] = sconf[p]; row[5] = P.scconf[o0 + p]; row[6] = (float)lt;
            slabel[p] = -1;                                     // the pivot is always retired
            s_p = mn;
        }
        __syncthreads();
        p = s_p; ++K;
        __syncthreads();
    }
    if (tid == 0) P.out_counts[b] = K;
}

constexpr int kCap = 16384;            // candidates per image sorted / bucketed in LDS (8 B each = 128 KB)
constexpr int kMaxCls = 1024;
constexpr int kRegE = 4;               // register-resident class lists: up to 64 * kRegE entries per wave

__device__ __forceinline__ void bitonic_sort_lds(float* key, unsigned* idx, int npad) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int lj = __ffs(k) - 2; lj >= 0; --lj) {            // j = 2^lj: shifts and masks instead of run-time divisions
            const int j = 1 << lj;
            for (int t = threadIdx.x; t < npad / 2; t += kImgT) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                const bool up = ((i & k) == 0);
                const float ki = key[i], kl = key[l];
                const unsigned ii = idx[i], il = idx[l];
                const bool sw = up ? before(kl, il, ki, ii) : before(ki, ii, kl, il);
                if (sw) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
            }
            __syncthreads();
        }
    }
}

// Image kernel, class-parallel.  `invalid = IoU > thr & SAME LABEL` (YOLOV4.py:271-274) means a pivot only ever removes
// boxes of its own class, so the reference's single sequential loop is the rank-ordered MERGE of independent per-class loops:
//   1. stable compaction of the valid (score, index) pairs into LDS, LDS bitonic sort (score desc, index asc) -> global rank
//   2. gather into rank order (global SoA), labels to LDS, class histogram + scan, per-class rank lists in LDS (ascending rank)
//   3. one WAVE per class runs that class's pivot loop -- lists of <= 256 entries live in registers (no memory in the loop),
//      longer ones walk LDS/L2 -- writes the merged row at the pivot's rank and raises its flag
//   4. block scan over the flags compacts the rows in rank order = the reference's output order.
// Sums are taken in a fixed order (lane partials in entry order, then one shuffle tree): reproducible run to run.
__global__ __launch_bounds__(kImgT) void yolo_nms_image_kernel(const YParams P) {
    extern __shared__ unsigned char s_dyn[];
    __shared__ int s_cnt[kMaxCls];
    __shared__ int s_off[kMaxCls];
    __shared__ int s_wtot[kImgT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nc = P.no - 5;
    const float* key = P.key + (size_t)b * P.Npad;
    const unsigned* idx = P.idx + (size_t)b * P.Npad;
    int nv = 0;                                                // valid candidates of this image (YOLOV4.py:258)
    {
        int c = 0;
        for (int n = tid; n < P.N; n += kImgT) c += idx[n] != 0xffffffffu;
        c = wave_sum_i(c);
        if (lane == 0) s_wtot[wv] = c;
        __syncthreads();
        for (int w = 0; w < kImgT / 64; ++w) nv += s_wtot[w];
        __syncthreads();
    }
    if (nv == 0) {                                             // "continue": output[image_i] stays None
        if (tid == 0) P.out_counts[b] = -1;
        return;
    }
    if (nv > kCap || nc > kMaxCls) { yolo_nms_image_serial(P, nv); return; }
    float* s_key = reinterpret_cast<float*>(s_dyn);
    unsigned* s_idx = reinterpret_cast<unsigned*>(s_dyn + 4 * kCap);
    float* out = P.out + (size_t)b * P.N * 7;
    const size_t o0 = (size_t)b * P.N;
    const unsigned long long lt_mask = lane ? (~0ull >> (64 - lane)) : 0ull;
    // ---- 1. compaction (index order) + sort
    int base = 0;
    for (int n0 = 0; n0 < P.N; n0 += kImgT) {
        const int n = n0 + tid;
        const bool v = n < P.N && idx[n] != 0xffffffffu;
        const unsigned long long m = __ballot(v);
        if (lane == 0) s_wtot[wv] = __popcll(m);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < kImgT / 64; ++w) { const int c = s_wtot[w]; if (w < wv) woff += c; tot += c; }
        if (v) { const int d = base + woff + __popcll(m & lt_mask); s_key[d] = key[n]; s_idx[d] = (unsigned)n; }
        base += tot;
        __syncthreads();
    }
    int npad = 2; while (npad < nv) npad <<= 1;
    for (int r = nv + tid; r < npad; r += kImgT) { s_key[r] = -INFINITY; s_idx[r] = 0xffffffffu; }
    __syncthreads();
    bitonic_sort_lds(s_key, s_idx, npad);
    // ---- 2. gather into rank order; labels, histogram, class offsets, class lists
    for (int c = tid; c < kMaxCls; c += kImgT) s_cnt[c] = 0;
    for (int r = tid; r < nv; r += kImgT) {
        const unsigned n = s_idx[r];
        P.sbox[o0 + r] = P.box[o0 + n]; P.sconf[o0 + r] = P.conf[o0 + n];
        P.scconf[o0 + r] = P.cconf[o0 + n]; P.slabel[o0 + r] = P.label[o0 + n];
    }
    __syncthreads();                                            // s_key / s_idx are dead from here: the LDS block is re-cut
    unsigned short* s_lab = reinterpret_cast<unsigned short*>(s_dyn);              // [kCap]  2 B
    unsigned char* s_flag = reinterpret_cast<unsigned char*>(s_dyn + 2 * kCap);    // [kCap]  1 B
    int* s_list = reinterpret_cast<int*>(s_dyn + 4 * kCap);                        // [kCap]  4 B
    for (int r = tid; r < nv; r += kImgT) {
        const int l = P.slabel[o0 + r];                         // written by this very thread above
        s_lab[r] = (unsigned short)l; s_flag[r] = 0;
        atomicAdd(&s_cnt[l], 1);
    }
    __syncthreads();
    {   // exclusive scan of the class counts (kMaxCls == kImgT: one class per thread)
        const int c = s_cnt[tid];
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_wtot[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += s_wtot[w];
        s_off[tid] = woff + inc - c;
        __syncthreads();
    }
    // ---- 3. per-class pivot loops, one wave per class
    const float4* sbox = P.sbox + o0; const float* sconf = P.sconf + o0; const float* scconf = P.scconf + o0;
    for (int c = wv; c < nc; c += kImgT / 64) {
        const int m = s_cnt[c];
        if (m == 0) continue;
        int* list = s_list + s_off[c];
        int run = 0;
        for (int r0 = 0; r0 < nv && run < m; r0 += 64) {        // ranks of this class, ascending
            const int r = r0 + lane;
            const bool hit = r < nv && s_lab[r] == (unsigned short)c;
            const unsigned long long mk = __ballot(hit);
            if (hit) list[run + __popcll(mk & lt_mask)] = r;
            run += __popcll(mk);
        }
        __threadfence_block();
        if (m <= 64 * kRegE) {
            float4 bx[kRegE]; float cf[kRegE], ccf[kRegE]; int rk[kRegE];
            unsigned long long am[kRegE];
#pragma unroll
            for (int u = 0; u < kRegE; ++u) {
                const int e = lane + 64 * u;
                const bool in = e < m;
                rk[u] = in ? list[e] : 0;
                bx[u] = in ? sbox[rk[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
                cf[u] = in ? sconf[rk[u]] : 0.f; ccf[u] = in ? scconf[rk[u]] : 0.f;
                am[u] = __ballot(in);
            }
            for (;;) {
                int pu = -1;
#pragma unroll
                for (int u = kRegE - 1; u >= 0; --u) if (am[u]) pu = u;
                if (pu < 0) break;
                float4 tb = make_float4(0.f, 0.f, 0.f, 0.f); float tcf = 0.f, tccf = 0.f; int trk = 0, pl = 0;
#pragma unroll
                for (int u = 0; u < kRegE; ++u) if (u == pu) {
                    pl = __ffsll((long long)am[u]) - 1;
                    tb.x = __shfl(bx[u].x, pl, 64); tb.y = __shfl(bx[u].y, pl, 64); tb.z = __shfl(bx[u].z, pl, 64); tb.w = __shfl(bx[u].w, pl, 64);
                    tcf = __shfl(cf[u], pl, 64); tccf = __shfl(ccf[u], pl, 64); trk = __shfl(rk[u], pl, 64);
                }
                Box4 top; top.x1 = tb.x; top.y1 = tb.y; top.x2 = tb.z; top.y2 = tb.w;
                float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int u = 0; u < kRegE; ++u) {
                    Box4 bj; bj.x1 = bx[u].x; bj.y1 = bx[u].y; bj.x2 = bx[u].z; bj.y2 = bx[u].w;
                    const bool inv = ((am[u] >> lane) & 1ull) && (iou_plus1(top, bj) > P.nms_thres);      // YOLOV4.py:271-274
                    if (inv) { const float w = cf[u]; acc[0] += w; acc[1] += w * bx[u].x; acc[2] += w * bx[u].y; acc[3] += w * bx[u].z; acc[4] += w * bx[u].w; }
                    am[u] &= ~__ballot(inv);
                }
#pragma unroll
                for (int u = 0; u < kRegE; ++u) if (u == pu) am[u] &= ~(1ull << pl);                       // the pivot is always retired
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = wave_sum(acc[i]);
                if (lane == 0) {
                    float* row = out + (size_t)trk * 7;                                                    // YOLOV4.py:275-279
                    row[0] = acc[1] / acc[0]; row[1] = acc[2] / acc[0]; row[2] = acc[3] / acc[0]; row[3] = acc[4] / acc[0];
                    row[4] = tcf; row[5] = tccf; row[6] = (float)c;
                    s_flag[trk] = 1;
                }
            }
        } else {
            int first = 0;
            for (;;) {
                int p = -1;
                for (int i0 = first; i0 < m; i0 += 64) {
                    const int i = i0 + lane;
                    const unsigned long long mk = __ballot(i < m && list[i] >= 0);
                    if (mk) { p = i0 + __ffsll((long long)mk) - 1; break; }
                }
                if (p < 0) break;
                const int rp = list[p];
                const float4 tb = sbox[rp];
                Box4 top; top.x1 = tb.x; top.y1 = tb.y; top.x2 = tb.z; top.y2 = tb.w;
                float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
                for (int i = p + lane; i < m; i += 64) {
                    const int r = list[i];
                    if (r < 0) continue;
                    const float4 q = sbox[r];
                    Box4 bj; bj.x1 = q.x; bj.y1 = q.y; bj.x2 = q.z; bj.y2 = q.w;
                    if (iou_plus1(top, bj) > P.nms_thres) {
                        const float w = sconf[r];
                        acc[0] += w; acc[1] += w * q.x; acc[2] += w * q.y; acc[3] += w * q.z; acc[4] += w * q.w;
                        list[i] = ~r;
                    }
                }
                __threadfence_block();
                if (lane == 0 && list[p] >= 0) list[p] = ~rp;                                             // the pivot is always retired
#pragma unroll
                for (int i = 0; i < 5; ++i) acc[i] = wave_sum(acc[i]);
                if (lane == 0) {
                    float* row = out + (size_t)rp * 7;
                    row[0] = acc[1] / acc[0]; row[1] = acc[2] / acc[0]; row[2] = acc[3] / acc[0]; row[3] = acc[4] / acc[0];
                    row[4] = sconf[rp]; row[5] = scconf[rp]; row[6] = (float)c;
                    s_flag[rp] = 1;
                }
                __threadfence_block();
                first = p + 1;
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    // ---- 4. rows were written at their pivot's rank; compact them in rank order (destination <= source: chunked, read-then-write)
    int K = 0;
    for (int r0 = 0; r0 < nv; r0 += kImgT) {
        const int r = r0 + tid;
        const bool f = r < nv && s_flag[r];
        float row[7];
        if (f) { for (int i = 0; i < 7; ++i) row[i] = out[(size_t)r * 7 + i]; }
        const unsigned long long mk = __ballot(f);
        if (lane == 0) s_wtot[wv] = __popcll(mk);
        __syncthreads();
        int woff = 0, tot = 0;
        for (int w = 0; w < kImgT / 64; ++w) { const int cw = s_wtot[w]; if (w < wv) woff += cw; tot += cw; }
        if (f) { float* d = out + (size_t)(K + woff + __popcll(mk & lt_mask)) * 7; for (int i = 0; i < 7; ++i) d[i] = row[i]; }
        K += tot;
        __threadfence_block();
        __syncthreads();
    }
    if (tid == 0) P.out_counts[b] = K;
}

// =================================================================================================
// [r4] The same NMS split across the chip (networks with >= kSplitMinCls classes).  One workgroup per image left 240 of 256 CUs
// idle and spent most of its time in a 91-stage LDS bitonic sort of ~5 000 keys.  But a pivot only ever removes boxes of ITS
// class (YOLOV4.py:271-274), so no global order is needed before the loops -- only inside each class -- and the reference's output
// order (pivots in rank order) is a sort of the ~1 000 rows that survive:
//   bucket (1 WG / image)        LDS class histogram + scan, (score, candidate) pairs scattered into class-contiguous segments
//   class  (1 WG / image, class) sorts its segment by (score desc, candidate asc) -- <= 256 entries: one wave, LDS bitonic network,
//                                the pivot loop with the list in registers (the loop of yolo_nms_image_kernel, same sums in the same
//                                order); longer segments: rank by counting + a workgroup-wide sweep per pivot -- and appends each
//                                merged row at the front of the class's own slots and, with ONE integer add per class, its pivots'
//                                (score, candidate, slot) as a run of the image's compact kept list
//   order  (64 rows / WG)        the rank of every kept row's (score, candidate) among the image's kept rows, all pairs by ballot +
//                                popcount; copies the rows out
// (score desc, candidate asc) is a strict total order: the output does not depend on which class finished first.
constexpr int kSplitMinCls = 8;        // below this many classes segments are long and the single-workgroup kernel is the better shape
constexpr int kClsT = 256;
constexpr int kKtotPitch = 64;         // one counter per 256-B line: agent-scope atomics on one line are served one at a time

constexpr int kBkU = 16;              // candidates per thread held in registers between the two passes (16 384 per image; more: chunks)

__global__ __launch_bounds__(kImgT) void yolo_nms_bucket_kernel(const YParams P) {
    __shared__ int s_cnt[kMaxCls], s_off[kMaxCls], s_cur[kMaxCls];
    __shared__ int s_wtot[kImgT / 64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nc = P.no - 5;
    const float* key = P.key + (size_t)b * P.Npad;
    const unsigned* idx = P.idx + (size_t)b * P.Npad;
    const int* label = P.label + (size_t)b * P.N;
    for (int c = tid; c < kMaxCls; c += kImgT) { s_cnt[c] = 0; s_cur[c] = 0; }
    // every load of the first chunk is requested before anything is consumed: one trip to memory instead of one per iteration
    // (the loop with its LDS atomics in between took 16 us for 11 iterations); an invalid candidate carries label -1
    int lab[kBkU]; float ky[kBkU];
#pragma unroll
    for (int u = 0; u < kBkU; ++u) {
        const int n = tid + u * kImgT, nn = n < P.N ? n : P.N - 1;
        const unsigned id = idx[nn];
        const int l = label[nn];
        ky[u] = key[nn];
        lab[u] = (n < P.N && id != 0xffffffffu) ? l : -1;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < kBkU; ++u) if (lab[u] >= 0) atomicAdd(&s_cnt[lab[u]], 1);
    for (int n = tid + kBkU * kImgT; n < P.N; n += kImgT) if (idx[n] != 0xffffffffu) atomicAdd(&s_cnt[label[n]], 1);
    __syncthreads();
    {   // exclusive scan of the class counts (kMaxCls == kImgT: one class per thread)
        const int c = s_cnt[tid];
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
        if (lane == 63) s_wtot[wv] = inc;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += s_wtot[w];
        s_off[tid] = woff + inc - c;
        if (tid < nc) { P.cls_off[(size_t)b * kMaxCls + tid] = woff + inc - c; P.cls_cnt[(size_t)b * kMaxCls + tid] = c; }
        if (tid == kImgT - 1) {
            const int nv = woff + inc;
            P.nvalid[b] = nv; P.ktot[(size_t)b * kKtotPitch] = 0;
            if (nv == 0) P.out_counts[b] = -1;                  // "continue": output[image_i] stays None
        }
        __syncthreads();
    }
    float* ckey = P.ckey + (size_t)b * P.Npad;
    unsigned* cidx = P.cidx + (size_t)b * P.Npad;
#pragma unroll
    for (int u = 0; u < kBkU; ++u) if (lab[u] >= 0) {
        const int slot = s_off[lab[u]] + atomicAdd(&s_cur[lab[u]], 1);   // any order inside the segment: the class kernel sorts it
        ckey[slot] = ky[u]; cidx[slot] = (unsigned)(tid + u * kImgT);
    }
    for (int n = tid + kBkU * kImgT; n < P.N; n += kImgT) if (idx[n] != 0xffffffffu) {
        const int l = label[n];
        const int slot = s_off[l] + atomicAdd(&s_cur[l], 1);
        ckey[slot] = key[n]; cidx[slot] = (unsigned)n;
    }
}

// one merged row of image b at slot `slot` of krow
__device__ __forceinline__ void emit_row(const YParams& P, int b, int slot, const float (&acc)[5], float cf, float ccf, int c) {
    float* row = P.krow + ((size_t)b * P.N + slot) * 7;                                                    // YOLOV4.py:275-279
    row[0] = acc[1] / acc[0]; row[1] = acc[2] / acc[0]; row[2] = acc[3] / acc[0]; row[3] = acc[4] / acc[0];
    row[4] = cf; row[5] = ccf; row[6] = (float)c;
}

// Sum over the 64 lanes without LDS traffic: an inclusive scan inside each row of 16 lanes by four data-parallel-primitive adds
// (row_shr 1, 2, 4, 8; out-of-row sources read 0), then the four row totals (lanes 15, 31, 47, 63) added in row order.  A fixed order
// (reproducible); the total is wave-uniform.  (__shfl_down trees are ds_bpermute chains: ~40 % of a pivot's time with five sums.)
__device__ __forceinline__ float wave_total_dpp(float v) {
#define ODPL_DPP_ADD(ctrl) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), ctrl, 0xf, 0xf, true))
    ODPL_DPP_ADD(0x111); ODPL_DPP_ADD(0x112); ODPL_DPP_ADD(0x114); ODPL_DPP_ADD(0x118);
#undef ODPL_DPP_ADD
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 15)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 31));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 47)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 63));
    return ((r0 + r1) + r2) + r3;
}

__device__ __forceinline__ int wave_total_dpp_i(int v) {
#define ODPL_DPP_ADDI(ctrl) v += __builtin_amdgcn_update_dpp(0, v, ctrl, 0xf, 0xf, true)
    ODPL_DPP_ADDI(0x111); ODPL_DPP_ADDI(0x112); ODPL_DPP_ADDI(0x114); ODPL_DPP_ADDI(0x118);
#undef ODPL_DPP_ADDI
    return __builtin_amdgcn_readlane(v, 15) + __builtin_amdgcn_readlane(v, 31) + __builtin_amdgcn_readlane(v, 47) + __builtin_amdgcn_readlane(v, 63);
}

// A class of <= 64 candidates (the usual case with many classes): the pairwise decisions `IoU(+1) > thr` do not depend on the order of
// the loop, so the whole m x m matrix is evaluated up front, one row per lane and all rows in parallel (m steps of one IoU), and the
// sequential pivot loop works on 64-bit masks in scalar registers: pivot = first alive bit, invalid = its row & alive.  A pivot that
// absorbs nobody but itself -- most of them -- yields the row (w x) / w of its own box, computed for every lane beforehand and written
// after the loop, all of them at once; only a real merge takes the wave sums.  Same rows as class_pivots<1> (the +1 IoU is symmetric
// in its arguments: max / min and one commutative sum per term), in the same order.
__device__ __forceinline__ int class_pivots_small(const YParams& P, int b, int c, int m, int off, float* s_k, unsigned* s_i) {
    const int lane = threadIdx.x & 63;
    const size_t o0 = (size_t)b * P.N;
    const bool in = lane < m;
    const unsigned cn = in ? s_i[lane] : 0u;
    const float kk = in ? s_k[lane] : 0.f;
    const float4 bx = in ? P.box[o0 + cn] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float cf = in ? P.conf[o0 + cn] : 0.f, ccf = in ? P.cconf[o0 + cn] : 0.f;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the sorted list is in registers: s_k / s_i now collect the kept pivots
    __builtin_amdgcn_wave_barrier();
    Box4 me; me.x1 = bx.x; me.y1 = bx.y; me.x2 = bx.z; me.y2 = bx.w;
    unsigned long long row = 0ull;
    for (int j = 0; j < m; ++j) {                               // (uniform trip count)
#define ODPL_RLJ(x) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), j))
        Box4 o; o.x1 = ODPL_RLJ(bx.x); o.y1 = ODPL_RLJ(bx.y); o.x2 = ODPL_RLJ(bx.z); o.y2 = ODPL_RLJ(bx.w);
#undef ODPL_RLJ
        if (iou_plus1(o, me) > P.nms_thres) row |= 1ull << j;   // YOLOV4.py:271-274: this lane as the pivot, candidate j against it
    }
    // the row a pivot leaves when it is alone in its `invalid` set: (w * x) / w per coordinate (YOLOV4.py:275-279 on one box)
    const float solo0 = (cf * bx.x) / cf, solo1 = (cf * bx.y) / cf, solo2 = (cf * bx.z) / cf, solo3 = (cf * bx.w) / cf;
    const unsigned row_lo = (unsigned)row, row_hi = (unsigned)(row >> 32);
    unsigned long long alive = m >= 64 ? ~0ull : ((1ull << m) - 1ull);
    int my_slot = -1;                                           // >= 0: this lane's candidate was a pivot and kept row `my_slot`
    bool my_solo = false;
    int nrows = 0;
    while (alive) {
        const int p = __ffsll((long long)alive) - 1;
        const unsigned long long rp = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)row_hi, p) << 32) |
                                      (unsigned)__builtin_amdgcn_readlane((int)row_lo, p);
        const unsigned long long inv = rp & alive;
        const bool solo = inv == (1ull << p);
        if (lane == p) { my_slot = nrows; my_solo = solo; }
        if (!solo) {                                            // (uniform) a merge, or a box whose own IoU is not above the threshold
            const bool mine = (inv >> lane) & 1ull;
            float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
            if (mine) { acc[0] = cf; acc[1] = cf * bx.x; acc[2] = cf * bx.y; acc[3] = cf * bx.z; acc[4] = cf * bx.w; }
#pragma unroll
            for (int i = 0; i < 5; ++i) acc[i] = wave_total_dpp(acc[i]);
            const float tcf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, cf), p));
            const float tccf = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ccf), p));
            const float num = lane == 0 ? acc[1] : lane == 1 ? acc[2] : lane == 2 ? acc[3] : acc[4];
            const float v = lane < 4 ? num / acc[0] : lane == 4 ? tcf : lane == 5 ? tccf : (float)c;
            if (lane < 7) P.krow[(o0 + off + nrows) * 7 + lane] = v;
        }
        alive &= ~(inv | (1ull << p));                          // the pivot is always retired
        ++nrows;
    }
    if (my_slot >= 0) {
        if (my_solo) {
            float* r = P.krow + (o0 + off + my_slot) * 7;
            r[0] = solo0; r[1] = solo1; r[2] = solo2; r[3] = solo3; r[4] = cf; r[5] = ccf; r[6] = (float)c;
        }
        s_k[my_slot] = kk; s_i[my_slot] = cn;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int base = 0;
    if (lane == 0) base = atomicAdd(&P.ktot[(size_t)b * kKtotPitch], nrows);
    base = __builtin_amdgcn_readfirstlane(base);
    const size_t op = (size_t)b * P.Npad;
    if (lane < nrows) { P.kkey[op + base + lane] = s_k[lane]; P.kidx[op + base + lane] = s_i[lane]; P.ksrc[op + base + lane] = off + lane; }
    return nrows;
}

// the pivot loop of one class with its (sorted) list in RE x 64 register slots; returns the number of rows it kept
template <int RE>
__device__ __forceinline__ int class_pivots(const YParams& P, int b, int c, int m, int off, float* s_k, unsigned* s_i) {
    const int lane = threadIdx.x & 63;
    const size_t o0 = (size_t)b * P.N;
    float4 bx[RE]; float cf[RE], ccf[RE], kk[RE]; unsigned cn[RE];
    unsigned long long am[RE];
#pragma unroll
    for (int u = 0; u < RE; ++u) {
        const int e = lane + 64 * u;
        const bool in = e < m;
        cn[u] = in ? s_i[e] : 0u; kk[u] = in ? s_k[e] : 0.f;
        bx[u] = in ? P.box[o0 + cn[u]] : make_float4(0.f, 0.f, 0.f, 0.f);
        cf[u] = in ? P.conf[o0 + cn[u]] : 0.f; ccf[u] = in ? P.cconf[o0 + cn[u]] : 0.f;
        am[u] = __ballot(in);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the sorted list is in registers: s_k / s_i now collect the kept pivots
    __builtin_amdgcn_wave_barrier();
    int nrows = 0;
    for (;;) {
        int pu = -1;
#pragma unroll
        for (int u = RE - 1; u >= 0; --u) if (am[u]) pu = u;
        if (pu < 0) break;
        float4 tb = make_float4(0.f, 0.f, 0.f, 0.f); float tcf = 0.f, tccf = 0.f, tk = 0.f; unsigned tn = 0u; int pl = 0;
#pragma unroll
        for (int u = 0; u < RE; ++u) if (u == pu) {             // the pivot's lane is wave-uniform: v_readlane, no LDS crossbar
            pl = __ffsll((long long)am[u]) - 1;
#define ODPL_RL(x) __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), pl))
            tb.x = ODPL_RL(bx[u].x); tb.y = ODPL_RL(bx[u].y); tb.z = ODPL_RL(bx[u].z); tb.w = ODPL_RL(bx[u].w);
            tcf = ODPL_RL(cf[u]); tccf = ODPL_RL(ccf[u]); tk = ODPL_RL(kk[u]); tn = (unsigned)__builtin_amdgcn_readlane((int)cn[u], pl);
#undef ODPL_RL
        }
        Box4 top; top.x1 = tb.x; top.y1 = tb.y; top.x2 = tb.z; top.y2 = tb.w;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < RE; ++u) {
            Box4 bj; bj.x1 = bx[u].x; bj.y1 = bx[u].y; bj.x2 = bx[u].z; bj.y2 = bx[u].w;
            const bool inv = ((am[u] >> lane) & 1ull) && (iou_plus1(top, bj) > P.nms_thres);          // YOLOV4.py:271-274
            if (inv) { const float w = cf[u]; acc[0] += w; acc[1] += w * bx[u].x; acc[2] += w * bx[u].y; acc[3] += w * bx[u].z; acc[4] += w * bx[u].w; }
            am[u] &= ~__ballot(inv);
        }
#pragma unroll
        for (int u = 0; u < RE; ++u) if (u == pu) am[u] &= ~(1ull << pl);                           // the pivot is always retired
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = wave_total_dpp(acc[i]);
        {   // the row (YOLOV4.py:275-279): the sums are wave-uniform, so lanes 0..3 take one quotient each and lanes 0..6 one store each
            const float num = lane == 0 ? acc[1] : lane == 1 ? acc[2] : lane == 2 ? acc[3] : acc[4];
            const float v = lane < 4 ? num / acc[0] : lane == 4 ? tcf : lane == 5 ? tccf : (float)c;
            if (lane < 7) P.krow[(o0 + off + nrows) * 7 + lane] = v;
            if (lane == 0) { s_k[nrows] = tk; s_i[nrows] = tn; }
        }
        ++nrows;
    }
    // a run of the image's compact kept list for this class: one integer add, then the pairs from LDS
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    int base = 0;
    if (lane == 0) base = atomicAdd(&P.ktot[(size_t)b * kKtotPitch], nrows);
    base = __shfl(base, 0, 64);
    const size_t op = (size_t)b * P.Npad;
    for (int j = lane; j < nrows; j += 64) { P.kkey[op + base + j] = s_k[j]; P.kidx[op + base + j] = s_i[j]; P.ksrc[op + base + j] = off + j; }
    return nrows;
}

__global__ __launch_bounds__(kClsT) void yolo_nms_class_kernel(const YParams P) {
    __shared__ float s_k[64 * kRegE];
    __shared__ unsigned s_i[64 * kRegE];
    __shared__ float s_red[6 * (kClsT / 64)];
    __shared__ int s_min[kClsT / 64];
    __shared__ int s_p;
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int m = P.cls_cnt[(size_t)b * kMaxCls + c];
    if (m == 0) return;
    const int off = P.cls_off[(size_t)b * kMaxCls + c];
    int nrows = 0;                                              // (the long-segment path's count)
    const size_t o0 = (size_t)b * P.N, op = (size_t)b * P.Npad;
    const float* ckey = P.ckey + op + off;
    const unsigned* cidx = P.cidx + op + off;
    if (m <= 64 * kRegE) {
        if (wv != 0) return;                                    // one wave: LDS traffic of a single wave is ordered, no barriers needed
        int npad = 64; while (npad < m) npad <<= 1;
        for (int e = lane; e < npad; e += 64) { s_k[e] = e < m ? ckey[e] : -INFINITY; s_i[e] = e < m ? cidx[e] : 0xffffffffu; }
        __builtin_amdgcn_wave_barrier();
        for (int k = 2; k <= npad; k <<= 1)
            for (int lj = __ffs(k) - 2; lj >= 0; --lj) {        // j = 2^lj: shifts and masks (t / j with a run-time j is a 40-instruction division)
                const int j = 1 << lj;
                for (int t = lane; t < npad / 2; t += 64) {
                    const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                    const bool up = ((i & k) == 0);
                    const float ki = s_k[i], kl = s_k[l];
                    const unsigned ii = s_i[i], il = s_i[l];
                    const bool sw = up ? before(kl, il, ki, ii) : before(ki, ii, kl, il);
                    if (sw) { s_k[i] = kl; s_k[l] = ki; s_i[i] = il; s_i[l] = ii; }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        if (m <= 64) class_pivots_small(P, b, c, m, off, s_k, s_i);
        else if (m <= 128) class_pivots<2>(P, b, c, m, off, s_k, s_i);
        else class_pivots<kRegE>(P, b, c, m, off, s_k, s_i);
        return;
    }
    // ---- a long segment (a class that dominates its image): rank by counting into the image's sorted arrays, then one workgroup-wide
    // sweep per pivot.  key / idx (the prep kernel's unsorted pairs) are dead after the bucket launch: their rows [off, off + m)
    // receive this segment in order; sbox / sconf / scconf / slabel (alive flag) likewise.
    float* skey = P.key + op + off; unsigned* sidx = P.idx + op + off;
    float4* sbox = P.sbox + o0 + off; float* sconf = P.sconf + o0 + off; float* scconf = P.scconf + o0 + off; int* alive = P.slabel + o0 + off;
    for (int e = tid; e < m; e += kClsT) {
        const float k = ckey[e]; const unsigned n = cidx[e];
        int r = 0;
        for (int f = 0; f < m; ++f) r += before(ckey[f], cidx[f], k, n);          // (a strict total order: ranks are a permutation)
        skey[r] = k; sidx[r] = n;
        sbox[r] = P.box[o0 + n]; sconf[r] = P.conf[o0 + n]; scconf[r] = P.cconf[o0 + n]; alive[r] = 1;
    }
    __threadfence_block();
    __syncthreads();
    int p = 0;
    while (p < m) {
        const float4 tb = sbox[p];
        Box4 top; top.x1 = tb.x; top.y1 = tb.y; top.x2 = tb.z; top.y2 = tb.w;
        float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
        int nxt = INT_MAX;
        for (int j = p + tid; j < m; j += kClsT) {
            if (!alive[j]) continue;
            const float4 q = sbox[j];
            Box4 bj; bj.x1 = q.x; bj.y1 = q.y; bj.x2 = q.z; bj.y2 = q.w;
            if (iou_plus1(top, bj) > P.nms_thres) {                               // YOLOV4.py:271-277 (same label: same segment)
                const float w = sconf[j];
                acc[0] += w; acc[1] += w * q.x; acc[2] += w * q.y; acc[3] += w * q.z; acc[4] += w * q.w;
                alive[j] = 0;
            } else if (j != p) {
                nxt = min(nxt, j);
            }
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) acc[i] = wave_sum(acc[i]);
        for (int o = 32; o > 0; o >>= 1) nxt = min(nxt, __shfl_down(nxt, o, 64));
        if (lane == 0) { for (int i = 0; i < 5; ++i) s_red[wv * 6 + i] = acc[i]; s_min[wv] = nxt; }
        __syncthreads();
        if (tid == 0) {
            float a[5] = {0.f, 0.f, 0.f, 0.f, 0.f}; int mn = INT_MAX;
            for (int w = 0; w < kClsT / 64; ++w) { for (int i = 0; i < 5; ++i) a[i] += s_red[w * 6 + i]; mn = min(mn, s_min[w]); }
            emit_row(P, b, off + nrows, a, sconf[p], scconf[p], c);
            const int at = atomicAdd(&P.ktot[(size_t)b * kKtotPitch], 1);         // (rare path: one reservation per row)
            P.kkey[op + at] = skey[p]; P.kidx[op + at] = sidx[p]; P.ksrc[op + at] = off + nrows;
            alive[p] = 0;                                                         // the pivot is always retired
            s_p = mn;
        }
        ++nrows;
        __threadfence_block();
        __syncthreads();
        p = s_p;
        __syncthreads();
    }
}

// order: the reference's output order is the rank of each kept row's (score, candidate) among all kept rows of its image (a strict
// total order).  K ~ 1 000 rows per image: all pairs is 1.2 M comparisons per image spread over the chip -- cheaper than any sort here (a
// 66-stage LDS bitonic network over 2 048 entries took 54 us in one workgroup; per-class binary searches are chains of dependent
// LDS reads: 79 us; all pairs out of LDS, one read per step: 33 us; out of registers with the two-part comparison: 31 us, 23 of them
// the comparisons).  Workgroup = kOrdRows rows of one image, kOrdRows / 4 per wave: every lane holds kOrdS kept pairs in REGISTERS (one
// trip to memory per chunk) as ONE 64-bit integer each whose unsigned order is (score desc, candidate asc), so a row against 64 pairs
// is one compare + one add per lane (the lanes' counts are summed once per row at the end), no memory access inside the loops.
constexpr int kOrdT = 256;
constexpr int kOrdRows = 64;
constexpr int kOrdS = 16;              // kept pairs per lane per chunk (1 024 pairs per chunk)

// a < b  <=>  before(ka, ia, kb, ib) for finite scores: descending score = ascending complement of the order-preserving bits
// (-0.0 is folded onto +0.0, which `before` treats as equal), candidate index in the low half
__device__ __forceinline__ unsigned long long order_key(float k, unsigned i) {
    const unsigned u = __float_as_uint(k + 0.0f);
    const unsigned asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)(~asc) << 32) | i;
}

__global__ __launch_bounds__(kOrdT) void yolo_nms_order_kernel(const YParams P) {
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (P.nvalid[b] == 0) return;                               // out_counts[b] = -1 was written by the bucket launch
    const int K = P.ktot[(size_t)b * kKtotPitch];
    if (blockIdx.x == 0 && tid == 0) P.out_counts[b] = K;
    constexpr int RPW = kOrdRows / (kOrdT / 64);                // rows per wave
    const int r0 = blockIdx.x * kOrdRows + wv * RPW;
    if (r0 >= K) return;                                        // the grid is sized for every candidate surviving
    const size_t o0 = (size_t)b * P.N, op = (size_t)b * P.Npad;
    const float* __restrict__ kkey = P.kkey + op; const unsigned* __restrict__ kidx = P.kidx + op;
    unsigned long long rkey[RPW]; int rank[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) { const int r = min(r0 + q, K - 1); rkey[q] = order_key(kkey[r], kidx[r]); rank[q] = 0; }      // (wave-uniform)
    for (int c0 = 0; c0 < K; c0 += 64 * kOrdS) {
        unsigned long long ek[kOrdS];
#pragma unroll
        for (int u = 0; u < kOrdS; ++u) {                      // pairs past the end: the largest key is in front of nothing
            const int e = c0 + u * 64 + lane;
            ek[u] = e < K ? order_key(kkey[e], kidx[e]) : ~0ull;
        }
#pragma unroll
        for (int u = 0; u < kOrdS; ++u) {
            if (c0 + u * 64 >= K) break;                        // (uniform)
#pragma unroll
            for (int q = 0; q < RPW; ++q) rank[q] += ek[u] < rkey[q];      // per lane: compare + add-with-carry, no scalar round trip
        }
    }
#pragma unroll
    for (int q = 0; q < RPW; ++q) rank[q] = wave_total_dpp_i(rank[q]);
    // the rows: every slot index first, then every value, then the stores (three trips for the wave's rows, not two per row)
    const int* __restrict__ ksrc = P.ksrc + op;
    const float* __restrict__ krow = P.krow + o0 * 7;
    float* __restrict__ out = P.out + o0 * 7;
    int src[RPW]; float val[RPW];
#pragma unroll
    for (int q = 0; q < RPW; ++q) src[q] = ksrc[min(r0 + q, K - 1)];
#pragma unroll
    for (int q = 0; q < RPW; ++q) val[q] = lane < 7 ? krow[(size_t)src[q] * 7 + lane] : 0.f;
#pragma unroll
    for (int q = 0; q < RPW; ++q) if (r0 + q < K && lane < 7) out[(size_t)rank[q] * 7 + lane] = val[q];
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int next_pow2(int v) { int p = 1; while (p < v) p <<= 1; return p; }

struct YPlan { size_t box, conf, cconf, label, sbox, sconf, scconf, slabel, key, idx, cls_off, cls_cnt, nvalid, ktot, ksrc, ckey, cidx, kkey, kidx, krow, total; int N, Npad; };

bool make_yplan(const odpl_yolo_nms_desc_t* d, YPlan* p) {
    if (!d || d->nscales < 1 || d->nscales > ODPL_MAX_SCALES || d->B < 1 || d->A < 1 || d->no < 6) return false;
    long long N = 0;
    for (int i = 0; i < d->nscales; ++i) { if (d->G[i] < 1 || !d->head[i]) return false; N += (long long)d->A * d->G[i] * d->G[i]; }
    if (N > (1 << 24)) return false;
    p->N = (int)N; p->Npad = next_pow2((int)N); if (p->Npad < 2) p->Npad = 2;
    const size_t BN = (size_t)d->B * N, BP = (size_t)d->B * p->Npad;
    size_t off = 0;
    p->box = off;    off += align_up(BN * 16, 256);
    p->conf = off;   off += align_up(BN * 4, 256);
    p->cconf = off;  off += align_up(BN * 4, 256);
    p->label = off;  off += align_up(BN * 4, 256);
    p->sbox = off;   off += align_up(BN * 16, 256);
    p->sconf = off;  off += align_up(BN * 4, 256);
    p->scconf = off; off += align_up(BN * 4, 256);
    p->slabel = off; off += align_up(BN * 4, 256);
    p->key = off;    off += align_up(BP * 4, 256);
    p->idx = off;    off += align_up(BP * 4, 256);
    p->cls_off = off; off += align_up((size_t)d->B * kMaxCls * 4, 256);
    p->cls_cnt = off; off += align_up((size_t)d->B * kMaxCls * 4, 256);
    p->nvalid = off; off += align_up((size_t)d->B * 4, 256);
    p->ktot = off;   off += align_up((size_t)d->B * kKtotPitch * 4, 256);
    p->ksrc = off;   off += align_up(BP * 4, 256);
    p->ckey = off;   off += align_up(BP * 4, 256);
    p->cidx = off;   off += align_up(BP * 4, 256);
    p->kkey = off;   off += align_up(BP * 4, 256);
    p->kidx = off;   off += align_up(BP * 4, 256);
    p->krow = off;   off += align_up(BN * 28, 256);
    p->total = off;
    return true;
}
}  // namespace

extern "C" size_t odpl_yolo_nms_workspace_bytes(const odpl_yolo_nms_desc_t* d) {
    YPlan p;
    return make_yplan(d, &p) ? p.total : 0;
}

extern "C" int odpl_yolo_nms(const odpl_yolo_nms_desc_t* d, void* workspace, size_t workspace_bytes,
                             float* out_dets, int32_t* out_counts, void* stream_) {
    YPlan pl;
    if (!make_yplan(d, &pl) || !workspace || workspace_bytes < pl.total || !out_dets || !out_counts) return ODPL_EINVAL;
    if ((uintptr_t)workspace & 15) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    YParams P;
    int first = 0;
    for (int i = 0; i < ODPL_MAX_SCALES; ++i) {
        P.head[i] = i < d->nscales ? d->head[i] : nullptr; P.G[i] = i < d->nscales ? d->G[i] : 1;
        P.first[i] = first; if (i < d->nscales) first += d->A * d->G[i] * d->G[i];
    }
    P.first[ODPL_MAX_SCALES] = first;
    for (int i = d->nscales; i < ODPL_MAX_SCALES; ++i) P.first[i] = first;
    P.nscales = d->nscales; P.B = d->B; P.A = d->A; P.no = d->no; P.N = pl.N; P.Npad = pl.Npad;
    P.conf_thres = d->conf_thres; P.nms_thres = d->nms_thres;
    P.box = (float4*)(ws + pl.box); P.conf = (float*)(ws + pl.conf); P.cconf = (float*)(ws + pl.cconf);
    P.label = (int*)(ws + pl.label);
    P.sbox = (float4*)(ws + pl.sbox); P.sconf = (float*)(ws + pl.sconf); P.scconf = (float*)(ws + pl.scconf);
    P.slabel = (int*)(ws + pl.slabel); P.key = (float*)(ws + pl.key); P.idx = (unsigned*)(ws + pl.idx);
    P.out = out_dets; P.out_counts = out_counts;
    P.cls_off = (int*)(ws + pl.cls_off); P.cls_cnt = (int*)(ws + pl.cls_cnt); P.nvalid = (int*)(ws + pl.nvalid); P.ktot = (int*)(ws + pl.ktot); P.ksrc = (int*)(ws + pl.ksrc);
    P.ckey = (float*)(ws + pl.ckey); P.cidx = (unsigned*)(ws + pl.cidx); P.kkey = (float*)(ws + pl.kkey); P.kidx = (unsigned*)(ws + pl.kidx);
    P.krow = (float*)(ws + pl.krow);
    prof_begin(ODPL_PROF_YOLO_NMS_PREP, stream);
    // class ranges per candidate split over 4 thread groups: measured 13.5 us at config 2 against 16.3 / 13.6 / 18.2 us for 1 / 2 / 8
    hipLaunchKernelGGL(yolo_nms_prep_kernel<4>, dim3((pl.Npad + 63) / 64, d->B), dim3(kPrepT), 0, stream, P);
    prof_end(ODPL_PROF_YOLO_NMS_PREP, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_YOLO_NMS_IMAGE, stream);
    static bool lds_attr_set[16] = {};                         // 128 KB of dynamic LDS needs the opt-in once per device
    int dev_id = 0;
    ODPL_HIP_TRY(hipGetDevice(&dev_id));
    if (dev_id < 0 || dev_id >= 16) return ODPL_EINVAL;
    if (!lds_attr_set[dev_id]) {
        ODPL_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(yolo_nms_image_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * kCap));
        lds_attr_set[dev_id] = true;
    }
    const int nc = d->no - 5;
    if (nc >= kSplitMinCls && nc <= kMaxCls && !(d->flags & ODPL_YOLO_NMS_ONE_WORKGROUP)) {
        hipLaunchKernelGGL(yolo_nms_bucket_kernel, dim3(d->B), dim3(kImgT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(yolo_nms_class_kernel, dim3(nc, d->B), dim3(kClsT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(yolo_nms_order_kernel, dim3((pl.N + kOrdRows - 1) / kOrdRows, d->B), dim3(kOrdT), 0, stream, P);
    } else {
        hipLaunchKernelGGL(yolo_nms_image_kernel, dim3(d->B), dim3(kImgT), 8 * kCap, stream, P);
    }
    prof_end(ODPL_PROF_YOLO_NMS_IMAGE, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

// =================================================================================================
// (2) RetinaNet / SSD greedy top-k NMS
namespace {
constexpr int kRows = 64;             // anchors per LDS sub-tile in the prep kernel (4 lanes scan one row)
constexpr int kPrepLoads = 5;         // 16-B loads in flight per lane while a tile is staged
constexpr int kMaxTopk = 1024;
constexpr int kKeyRegs = 48;           // candidate keys per thread held in registers during the top-k bisection (49 152 candidates)

struct GParams {
    const float* loc; const float* cls; const float* anchors;
    int B, An, C, topk;
    float nms_thresh, class_thresh;
    float* score; int* label;          // [B*An] all anchors
    float* ckey; int* cidx;            // [B*An] candidates (compacted, index order)
    unsigned magic_c4, magic_c;        // fastdiv16 magics of C/4 and C (tile-local indices < 2^16)
    int* tile_cnt; int ntiles;         // [B*ntiles] candidates per kRows-anchor tile (written by the prep kernel)
    int* ncand;                        // [B] candidates per image (scan launch)
    float* out; int* out_counts;
};

__device__ __forceinline__ Box4 decode_anchor(const float* loc4, const float* an4) {
    // RetinaNet.py:121-126: xy = loc_xy * a_wh + a_xy ; wh = exp(loc_wh) * a_wh ; box = (xy - wh/2, xy + wh/2)
    const float x = loc4[0] * an4[2] + an4[0], y = loc4[1] * an4[3] + an4[1];
    const float w = expf(loc4[2]) * an4[2], h = expf(loc4[3]) * an4[3];
    return corners_of(x, y, w, h);
}

// Scores and labels of one tile of kRows anchors: torch.max(sigmoid(cls), 1) (RetinaNet.py:128).
//
// The kernel streams 4*C bytes per anchor and is bounded by HBM only while a wave spends < ~500 VALU issues per 20-KB tile (a
// wave64 instruction occupies its SIMD for 4 cycles; tools/probe_rowmax.hip: 49 us with a bare row maximum, 99 us with 20
// exact sigmoids per lane).  So the sigmoid is evaluated on few elements.  With m the row's largest logit, an element
// l < m - 1/16 has sigmoid(l) STRICTLY below sigmoid(m) once both are rounded, provided l <= 8 (relative gap >=
// (1 - sigmoid(8)) / 16 = 2e-5, against a few 6e-8 ulps of evaluation error; above 8 neighbours can round to the same float, up
// to 1.0f) and m >= -80 (below, 1 + exp(-l) overflows towards a sigmoid of 0 for everything).  Such elements can neither win
// nor tie; all the others -- NaNs included, `!(l < thr)` -- are evaluated exactly, in index order.  Lanes walk a bit mask of
// their survivors: a wave runs max-over-lanes (about 2) sigmoids instead of C/4.
__device__ __forceinline__ float prep_threshold(float m) { return !(m >= -80.0f) ? -INFINITY : fminf(m - 0.0625f, 8.0f); }
__device__ __forceinline__ void prep_combine_and_store(const GParams& P, int b, int a0, int r, int q, int nr, float best, int bi) {
    // combine the 4 lanes of a row: larger value, then smaller index (torch.max(1): first index on ties)
    for (int o = 1; o < 4; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if ((ob != ob && (best == best || oi < bi)) || (best == best && (ob > best || (ob == best && oi < bi)))) { best = ob; bi = oi; }   // NaN wins, first one
    }
    if (r < nr && q == 0) {
        const size_t o = (size_t)b * P.An + a0 + r;
        P.score[o] = best; P.label[o] = bi;
    }
    // candidates of this tile (RetinaNet.py:129: score > class_thresh): the image kernel scans these counts instead of
    // sweeping all An scores behind workgroup barriers
    __shared__ int s_tc[4];
    const unsigned long long pm = __ballot(r < nr && q == 0 && best > P.class_thresh);
    if ((threadIdx.x & 63) == 0) s_tc[threadIdx.x >> 6] = __popcll(pm);
    __syncthreads();
    if (threadIdx.x == 0) P.tile_cnt[(size_t)b * P.ntiles + blockIdx.x] = s_tc[0] + s_tc[1] + s_tc[2] + s_tc[3];
}

// Class count known at compile time and a multiple of 16 (80: COCO, the BASELINE configs).  The tile is a plain contiguous copy
// into LDS (pitch C, no padding, no index arithmetic); lane q of a row owns the contiguous quarter [q*C/4, (q+1)*C/4) and reads
// it with 16-B LDS loads -- conflict-free without padding: the 16 lanes of one LDS pass start at the 16-B groups
// 20 r + 5 q (mod 16) = 4 r + 5 q, all distinct for r, q in 0..3 -- and keeps it in registers for both scans.
template <int CT>
__global__ __launch_bounds__(256) void greedy_nms_prep_ct_kernel(const GParams P) {
    static_assert(CT % 16 == 0 && CT / 4 <= 32, "quarter rows of whole 16-B vectors, one 32-bit survivor mask per lane");
    constexpr int C4 = CT / 4, PER = CT / 4, V = PER / 4, N4 = kRows * C4, U = (N4 + 255) / 256;
    extern __shared__ __align__(16) float s_tile[];         // kRows * CT
    const int b = blockIdx.y, tid = threadIdx.x;
    const int a0 = blockIdx.x * kRows;
    const int nr = (P.An - a0) < kRows ? (P.An - a0) : kRows;
    const float4* src4 = reinterpret_cast<const float4*>(P.cls + ((size_t)b * P.An + a0) * CT);
    float4* s4 = reinterpret_cast<float4*>(s_tile);
    const int n4 = nr * C4;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = (tid + u * 256) < n4 ? src4[tid + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int u = 0; u < U; ++u) if (tid + u * 256 < N4) s4[tid + u * 256] = v[u];
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    float l[PER];
#pragma unroll
    for (int i = 0; i < V; ++i) { const float4 t = s4[r * C4 + q * V + i]; l[4 * i] = t.x; l[4 * i + 1] = t.y; l[4 * i + 2] = t.z; l[4 * i + 3] = t.w; }
    float m = -INFINITY;
#pragma unroll
    for (int j = 0; j < PER; ++j) m = fmaxf(m, l[j]);                      // fmaxf skips NaNs
    m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
    const float thr = prep_threshold(m);
    unsigned need = 0u;
#pragma unroll
    for (int j = 0; j < PER; ++j) need |= (unsigned)(!(l[j] < thr)) << j;
    float best = -1.f; int bi = 0x7fffffff;                 // sigmoid > 0 > -1
    const float* mine = s_tile + r * CT + q * PER;
    while (need) {
        const int j = __builtin_ctz(need);
        need &= need - 1u;
        const float sg = sigmoidf_(mine[j]);
        if (sg > best || (sg != sg && best == best)) { best = sg; bi = q * PER + j; }
    }
    prep_combine_and_store(P, b, a0, r, q, nr, best, bi);
}

// Any class count: lanes q, q+4, ... of a row (pitch C+4 / C+1: conflict-free 4-B LDS reads), two scans of the LDS row.
__global__ __launch_bounds__(256) void greedy_nms_prep_kernel(const GParams P) {
    extern __shared__ __align__(16) float s_tile[];         // kRows * ld
    const int b = blockIdx.y, tid = threadIdx.x, C = P.C;
    const bool vec = (C & 3) == 0;
    const int ld = vec ? C + 4 : C + 1;                     // C+4: rows stay 16-B aligned
    const int a0 = blockIdx.x * kRows;
    const int nr = (P.An - a0) < kRows ? (P.An - a0) : kRows;
    const float* src = P.cls + ((size_t)b * P.An + a0) * C;
    const int n = nr * C;
    if (vec) {                                              // the 64 x C tile is contiguous: 16-B loads, kPrepLoads in flight per lane
        const float4* src4 = reinterpret_cast<const float4*>(src);
        const int n4 = n >> 2, c4 = C >> 2;
        for (int e = tid; e < n4; e += kPrepLoads * 256) {
            float4 v[kPrepLoads];
#pragma unroll
            for (int u = 0; u < kPrepLoads; ++u) v[u] = (e + u * 256) < n4 ? src4[e + u * 256] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < kPrepLoads; ++u) {
                const int ee = e + u * 256;
                if (ee < n4) { const int r = fastdiv16(ee, P.magic_c4); *reinterpret_cast<float4*>(&s_tile[r * ld + 4 * (ee - r * c4)]) = v[u]; }
            }
        }
    } else {
        for (int e = tid; e < n; e += 256) { const int r = fastdiv16(e, P.magic_c); s_tile[r * ld + (e - r * C)] = src[e]; }
    }
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    float best = -1.f; int bi = 0x7fffffff;                 // sigmoid > 0 > -1
    const float* row = s_tile + (r < nr ? r : 0) * ld;
    float m = -INFINITY;
    for (int k = q; k < C; k += 4) m = fmaxf(m, row[k]);                   // fmaxf skips NaNs
    m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
    const float thr = prep_threshold(m);
    unsigned long long need = 0ull;                                         // C <= 236: at most 59 elements per lane
    int j = 0;
    for (int k = q; k < C; k += 4, ++j) need |= (unsigned long long)(!(row[k] < thr)) << j;
    while (need) {
        const int k = q + 4 * __builtin_ctzll(need);
        need &= need - 1ull;
        const float sg = sigmoidf_(row[k]);
        if (sg > best || (sg != sg && best == best)) { best = sg; bi = k; }
    }
    prep_combine_and_store(P, b, a0, r, q, nr, best, bi);
}

// [r4] The per-image part as three launches.  One 1024-thread workgroup per image used to do everything -- four workgroups of a
// 256-CU chip at config 4 (166 us): 50 us walking 3 136 tiles eight at a time, 31 bisection rounds of two barriers each for the
// k-th largest score, up to 100 sequential pivots of ~0.4 us.  Now:
//   scan    (1 WG / image)     block scan of the prep kernel's per-tile counts -> tile offsets, the number of candidates
//   scatter (ntiles / 32 WGs)  the stable compaction itself, spread over the chip: wave = 8 tiles, ballot + popcount positions
//   final   (1 WG / image)     k-th largest score by a three-level radix select on the float bits with the keys in registers
//                              (an LDS histogram + one suffix scan per level instead of 31 rounds), the same gather / LDS sort /
//                              box decode as before, then the greedy loop on an overlap MATRIX: `ovr(i, j) > thr` does not depend on
//                              the order of the loop, so every lane evaluates its candidates' rows up front, all in parallel, and
//                              the sequential part works on bit masks in scalar registers (topk <= 128; longer lists keep the
//                              sweep).  Same kept sets, same output quirks.
__global__ __launch_bounds__(kImgT) void greedy_nms_scan_kernel(const GParams P) {
    __shared__ int s_wcnt[kImgT / 64];
    __shared__ int s_off;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int* tcnt = P.tile_cnt + (size_t)b * P.ntiles;
    if (tid == 0) s_off = 0;
    __syncthreads();
    for (int base = 0; base < P.ntiles; base += kImgT) {
        const int t = base + tid;
        const int c = t < P.ntiles ? tcnt[t] : 0;
        int inc = c;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) s_wcnt[wv] = inc;
        __syncthreads();
        int off = s_off, tot = 0;
        for (int w = 0; w < kImgT / 64; ++w) { if (w < wv) off += s_wcnt[w]; tot += s_wcnt[w]; }
        if (t < P.ntiles) tcnt[t] = c ? ((off + inc - c) | (c << 24)) : -1;      // offset (24 bits) | count (<= 64); -1 = empty tile
        __syncthreads();
        if (tid == 0) s_off += tot;
        __syncthreads();
    }
    if (tid == 0) P.ncand[b] = s_off;
}

constexpr int kScT = 256, kScTU = 8;                 // scatter: 4 waves x 8 tiles per workgroup

__global__ __launch_bounds__(kScT) void greedy_nms_scatter_kernel(const GParams P) {
    const int b = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const size_t o0 = (size_t)b * P.An;
    const float* score = P.score + o0;
    float* ckey = P.ckey + o0; int* cidx = P.cidx + o0;
    const int* tcnt = P.tile_cnt + (size_t)b * P.ntiles;
    const int t0 = (blockIdx.x * (kScT / 64) + wv) * kScTU;
    int oc[kScTU]; float sc[kScTU];
#pragma unroll
    for (int u = 0; u < kScTU; ++u) {                // independent loads first
        const int t = t0 + u;
        oc[u] = t < P.ntiles ? tcnt[t] : -1;
        const int a = t * kRows + lane;
        sc[u] = (t < P.ntiles && a < P.An) ? score[a] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < kScTU; ++u) {
        if (oc[u] < 0) continue;
        const int a = (t0 + u) * kRows + lane;
        const bool f = a < P.An && sc[u] > P.class_thresh;                        // RetinaNet.py:129-130
        const unsigned long long mk = __ballot(f);
        if (f) { const int pos = (oc[u] & 0xffffff) + __popcll(mk & ((1ull << lane) - 1ull)); ckey[pos] = sc[u]; cidx[pos] = a; }
    }
}

constexpr int kSelBins = 2048;                       // radix select: bins per level (two per thread)
constexpr int kListCap = 1024;                       // the short top-k: keys at or above the bound that fit the LDS list

#ifdef ODPL_NMS_TRACE
__device__ unsigned long long* g_nms_trace = nullptr;                      // [B][16] wall-clock stamps (100 MHz) of greedy_nms_image_kernel
#define NTRACE(slot) do { if (g_nms_trace && threadIdx.x == 0) g_nms_trace[(size_t)blockIdx.x * 16 + (slot)] = wall_clock64(); } while (0)
#else
#define NTRACE(slot) do { } while (0)
#endif

__global__ __launch_bounds__(kImgT) void greedy_nms_image_kernel(const GParams P) {
    __shared__ int s_wcnt[kImgT / 64], s_wcnt2[kImgT / 64];
    __shared__ int s_hist[kSelBins];
    __shared__ int s_sel[2];
    __shared__ float s_key[kMaxTopk]; __shared__ int s_pos[kMaxTopk];
    __shared__ float4 s_box[kMaxTopk]; __shared__ float s_area[kMaxTopk]; __shared__ int s_alive[kMaxTopk];
    __shared__ int s_keep[kMaxTopk];
    __shared__ int s_p;
    __shared__ int s_rank[128];
    __shared__ unsigned s_lkey[kListCap]; __shared__ int s_lidx[kListCap], s_la[kListCap], s_rank2[kListCap];
    __shared__ unsigned long long s_rows[128][2];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const size_t o0 = (size_t)b * P.An;
    float* ckey = P.ckey + o0; int* cidx = P.cidx + o0;
    NTRACE(0);
    const int n = P.ncand[b];
    float* out = P.out + (size_t)b * P.topk * 7;
    if (n <= 1) {            // 0: nothing; 1: the reference raises IndexError (0-dim squeeze) -- reported via counts
        if (tid == 0) { P.out_counts[b * 2] = 0; P.out_counts[b * 2 + 1] = n; }
        return;
    }
    // ---- top-k by score (RetinaNet.py:140-141): the k-th largest key T by radix select on the float bits (scores are positive
    //      finite floats: the bits order like the values).  Up to kKeyRegs * 1024 candidate keys are held in registers, so a level is
    //      one LDS-histogram pass + one suffix scan; more candidates than that re-read the keys from L2.
    int m = n < P.topk ? n : P.topk;
    unsigned T = 0;
    unsigned kreg[kKeyRegs];
    const bool in_regs = n <= kKeyRegs * kImgT;
    if (in_regs) {
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) { const int i = tid + j * kImgT; kreg[j] = i < n ? __float_as_uint(ckey[i]) : 0u; }
    }
    NTRACE(1);
    // ---- the short way to the top m <= 128 of keys held in registers.  The m-th largest of the THREADS' maxima is a lower bound L of
    //      the m-th largest key (m threads hold a key >= it), and only a few hundred of tens of thousands of keys reach it: one small
    //      histogram (one value per thread) finds L, the keys >= L go to an LDS list, and the list is ordered by COUNTING -- entry e
    //      counts the entries that come before it in (score desc, position asc), a strict total order, so the counts are the final
    //      places and places < m are the survivors: selection, tie rule (index order) and sort in one pass, with the entries' decode
    //      loads (candidate -> anchor -> offsets + anchor box) in flight under it.  A list that does not fit takes the general way.
    bool fast = false;
    if (in_regs && n > P.topk && m <= 128) {
        unsigned mx = 0u;
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) if (tid + j * kImgT < n) mx = kreg[j] > mx ? kreg[j] : mx;
        const unsigned lo_bits = P.class_thresh > 0.f ? __float_as_uint(P.class_thresh) : 0u;
        const unsigned hi_bits = lo_bits < 0x3f800000u ? 0x3f800000u : lo_bits;
        int span_log = 1; while (span_log < 31 && (1u << span_log) <= hi_bits - lo_bits) ++span_log;
        const int shift = span_log > 11 ? span_log - 11 : 0;
        s_hist[tid] = 0; s_hist[tid + kImgT] = 0;
        if (tid == 0) { s_sel[0] = -1; s_sel[1] = 0; }
        __syncthreads();
        const bool has_key = tid < n;
        const unsigned dmx = mx - lo_bits;
        const bool in_range = has_key && mx >= lo_bits && (dmx >> span_log) == 0u;
        if (in_range) atomicAdd(&s_hist[dmx >> shift], 1);       // a maximum beyond 1.0 (never, for sigmoids) is not counted: the bound only gets lower
        __syncthreads();
        const int h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
        int inc = h0 + h1;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_down(inc, o, 64); if (lane + o < 64) inc += u; }
        if (lane == 0) s_wcnt[wv] = inc;
        __syncthreads();
        int above = 0;
        for (int w = wv + 1; w < kImgT / 64; ++w) above += s_wcnt[w];
        const int S0 = above + inc, S1 = S0 - h0, S2 = S1 - h1;
        if (S1 >= m && S2 < m) s_sel[0] = 2 * tid + 1;
        else if (S0 >= m && S1 < m) s_sel[0] = 2 * tid;
        __syncthreads();
        const int bin = s_sel[0];
        __syncthreads();
        if (bin >= 0) {                                                   // (uniform) -1: the maxima did not fill the range as expected -> general way
            const unsigned L = lo_bits + ((unsigned)bin << shift);
            if (tid == 0) s_sel[0] = 0;
            __syncthreads();
            NTRACE(2);
            int hits = 0;                                                 // one cursor step per thread (most have none), then its hits in a row
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j) hits += (tid + j * kImgT < n && kreg[j] >= L);
            int at = hits ? atomicAdd(&s_sel[0], hits) : 0;
            if (hits) {
#pragma unroll
                for (int j = 0; j < kKeyRegs; ++j) {
                    const int i = tid + j * kImgT;
                    if (i < n && kreg[j] >= L) { if (at < kListCap) { s_lkey[at] = kreg[j]; s_lidx[at] = i; s_la[at] = cidx[i]; } ++at; }
                }
            }
            __syncthreads();
            const int c = s_sel[0];
            __syncthreads();
            NTRACE(3);
            if (c >= m && c <= kListCap) {                                // (uniform)
                fast = true;
                unsigned ke = 0u; int ie = 0;
                float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), an4 = lo4;
                if (tid < c) {                                            // entry tid: its key and its box loads (the anchor index is in LDS already)
                    ke = s_lkey[tid]; ie = s_lidx[tid];
                    const int a = s_la[tid];
                    lo4 = *reinterpret_cast<const float4*>(P.loc + (o0 + a) * 4);
                    an4 = *reinterpret_cast<const float4*>(P.anchors + (size_t)a * 4);
                }
                // counting, all threads: thread (e, g) counts the entries of slice g that come before entry e
                int cpad = 128; while (cpad < c) cpad <<= 1;
                const int G = kImgT / cpad, e = tid & (cpad - 1), g = tid / cpad, per = (c + G - 1) / G;
                s_rank2[tid] = 0;
                __syncthreads();
                if (e < c) {
                    const unsigned k2 = s_lkey[e]; const int i2 = s_lidx[e];
                    const int q0 = g * per, q1 = q0 + per < c ? q0 + per : c;
                    int cnt = 0;
                    for (int o = q0; o < q1; ++o) { const unsigned ko = s_lkey[o]; cnt += (ko > k2) || (ko == k2 && s_lidx[o] < i2); }
                    if (cnt) atomicAdd(&s_rank2[e], cnt);
                }
                __syncthreads();
                const int rank = tid < c ? s_rank2[tid] : 0;
                if (tid < c && rank < m) {
                    const float lo_[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, an_[4] = {an4.x, an4.y, an4.z, an4.w};
                    const Box4 bx = decode_anchor(lo_, an_);
                    s_key[rank] = __uint_as_float(ke); s_pos[rank] = ie;
                    s_box[rank] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
                    s_area[rank] = (bx.x2 - bx.x1 + 1.0f) * (bx.y2 - bx.y1 + 1.0f);
                }
                __syncthreads();
                NTRACE(4);
            }
        }
    }
    if (!fast) {
    if (n > P.topk) {
        // current range [base, base + 2^span_log), cut into bins of 2^shift; keys above it: c_above.  The first range is where the keys
        // can be -- above class_thresh, at most sigmoid's 1.0 -- so that the first histogram spreads over ~1 000 bins (the whole
        // positive range put 38 000 keys on ten LDS addresses: 17 us of same-address atomics)
        const unsigned lo_bits = P.class_thresh > 0.f ? __float_as_uint(P.class_thresh) : 0u;
        const unsigned hi_bits = lo_bits < 0x3f800000u ? 0x3f800000u : lo_bits;      // scores are sigmoids: 1 / (1 + e^-x) <= 1.0f
        unsigned base = lo_bits;
        int span_log = 1; while (span_log < 31 && (1u << span_log) <= hi_bits - lo_bits) ++span_log;
        int shift = span_log > 11 ? span_log - 11 : 0, c_above = 0;
        for (;;) {
            s_hist[tid] = 0; s_hist[tid + kImgT] = 0;
            __syncthreads();
            auto add = [&](unsigned u) { const unsigned d = u - base; if (u >= base && (d >> span_log) == 0u) atomicAdd(&s_hist[d >> shift], 1); };
            if (in_regs) {
#pragma unroll
                for (int j = 0; j < kKeyRegs; ++j) if (tid + j * kImgT < n) add(kreg[j]);
            } else {
                for (int i = tid; i < n; i += kImgT) add(__float_as_uint(ckey[i]));
            }
            __syncthreads();
            // suffix sums S(b) = hist[b] + hist[b + 1] + ...: thread t owns bins 2t, 2t + 1; waves scanned from the top
            const int h0 = s_hist[2 * tid], h1 = s_hist[2 * tid + 1];
            int inc = h0 + h1;                               // inclusive suffix over threads tid .. 1023
            for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_down(inc, o, 64); if (lane + o < 64) inc += u; }
            if (lane == 0) s_wcnt[wv] = inc;
            __syncthreads();
            int above = 0;
            for (int w = wv + 1; w < kImgT / 64; ++w) above += s_wcnt[w];
            const int S0 = c_above + above + inc;            // keys in bins >= 2t
            const int S1 = S0 - h0;                          // keys in bins >= 2t + 1
            const int S2 = S1 - h1;                          // keys in bins >= 2t + 2
            if (S1 >= m && S2 < m) { s_sel[0] = 2 * tid + 1; s_sel[1] = S2; }        // the bin that holds the m-th largest: exactly one
            else if (S0 >= m && S1 < m) { s_sel[0] = 2 * tid; s_sel[1] = S1; }
            __syncthreads();
            base += (unsigned)s_sel[0] << shift; c_above = s_sel[1];
            __syncthreads();
            if (shift == 0) break;
            span_log = shift; shift = shift > 11 ? shift - 11 : 0;
        }
        T = base;                                            // largest T with count(bits >= T) >= topk
    }
    // gather the selected candidates: every key > T, then keys == T in index order until m are taken.  The survivors are sorted
    // below, so their order here is free -- except among ties at T when not all of them fit.  With the keys in registers and every tie
    // fitting (the usual case: distinct scores), each thread appends its own hits through an LDS cursor: no pass over memory.
    bool gathered = false;
    if (in_regs) {
        int cg = 0, ce = 0;
#pragma unroll
        for (int j = 0; j < kKeyRegs; ++j) { const bool v = tid + j * kImgT < n; cg += v && kreg[j] > T; ce += v && kreg[j] == T; }
        cg = wave_sum_i(cg); ce = wave_sum_i(ce);
        if (tid == 0) { s_sel[0] = 0; s_sel[1] = 0; }
        __syncthreads();
        if (lane == 0) { if (cg) atomicAdd(&s_sel[0], cg); if (ce) atomicAdd(&s_sel[1], ce); }
        __syncthreads();
        const int n_gt = s_sel[0], n_eq = s_sel[1];
        __syncthreads();
        if (n <= P.topk || n_gt + n_eq == m) {               // every candidate at or above T is selected: no order to respect
            if (tid == 0) s_sel[0] = 0;
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kKeyRegs; ++j) {
                const int i = tid + j * kImgT;
                if (i < n && (n <= P.topk || kreg[j] >= T)) { const int pos = atomicAdd(&s_sel[0], 1); if (pos < m) { s_key[pos] = __uint_as_float(kreg[j]); s_pos[pos] = i; } }
            }
            __syncthreads();
            gathered = true;
        }
    }
    if (!gathered) {
        // more ties at T than fit (or keys not in registers): index order decides.  Each wave owns a contiguous slice of the
        // candidate list: count, one block scan, then write -- no barrier inside the sweeps.
        const int per = ((n + kImgT / 64 - 1) / (kImgT / 64) + 63) / 64 * 64;   // slice length, multiple of 64
        const int i0 = wv * per, i1 = (i0 + per) < n ? (i0 + per) : n;
        int cg = 0, ce = 0;
        for (int i = i0 + lane; i < i1; i += 64) { const unsigned u = __float_as_uint(ckey[i]); cg += (u > T); ce += (u == T); }
        cg = wave_sum_i(cg); ce = wave_sum_i(ce);
        __syncthreads();
        if (lane == 0) { s_wcnt[wv] = cg; s_wcnt2[wv] = ce; }
        __syncthreads();
        int og = 0, oe = 0, totg = 0;
        for (int w = 0; w < kImgT / 64; ++w) { if (w < wv) { og += s_wcnt[w]; oe += s_wcnt2[w]; } totg += s_wcnt[w]; }
        oe += totg;                                                              // the == T block follows all > T entries
        for (int base = i0; base < i1; base += 64) {
            const int i = base + lane;
            const unsigned u = i < i1 ? __float_as_uint(ckey[i]) : 0u;
            const bool fg = i < i1 && u > T, fe = i < i1 && u == T && n > P.topk;
            const unsigned long long mg = __ballot(fg), me = __ballot(fe);
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (fg) { const int pos = og + __popcll(mg & lt); if (pos < m) { s_key[pos] = __uint_as_float(u); s_pos[pos] = i; } }
            if (fe) { const int pos = oe + __popcll(me & lt); if (pos < m) { s_key[pos] = __uint_as_float(u); s_pos[pos] = i; } }
            og += __popcll(mg); oe += __popcll(me);
        }
        __syncthreads();
    }
    // ---- sort the m survivors by (score desc, filtered position asc) in LDS, then the candidate boxes (decode of the candidate's
    //      anchor) and areas (RetinaNet.py:133-139).  Up to 128 survivors (topk = 100) ONE wave does both without a workgroup barrier
    //      (LDS traffic of a single wave is ordered); longer lists take the whole workgroup.
    }                                                                     // !fast
    NTRACE(5);
    int mp = 1; while (mp < m) mp <<= 1;
    if (mp < 64) mp = 64;
    const bool one_wave = m <= 128;
    if (one_wave && !fast) {
        // Up to 128 survivors: every thread takes part.  Thread e < m owns survivor e: its decode loads (candidate -> anchor -> offsets and
        // anchor box: two dependent trips) go out first; meanwhile the order is found by COUNTING -- thread (e, g) counts the survivors of
        // block g that come before e ((score desc, position asc) is a strict total order: the counts are a permutation) -- and when the
        // rank is known the survivor lands at its place with its box.  One pass instead of a 28-step bitonic network + a decode behind it.
        const int e = tid & 127, g = tid >> 7;                           // 8 blocks of 16 survivors
        if (tid < 128) s_rank[tid] = 0;
        float my_key = 0.f; int my_pos = 0, my_a = 0;
        float4 lo4 = make_float4(0.f, 0.f, 0.f, 0.f), an4 = lo4;
        if (tid < m) {
            my_key = s_key[tid]; my_pos = s_pos[tid];
            my_a = cidx[my_pos];
            lo4 = *reinterpret_cast<const float4*>(P.loc + (o0 + my_a) * 4);
            an4 = *reinterpret_cast<const float4*>(P.anchors + (size_t)my_a * 4);
        }
        __syncthreads();
        if (e < m) {
            const float ke = s_key[e]; const unsigned pe = (unsigned)s_pos[e];
            int cnt = 0;
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                const int o = g * 16 + u;
                if (o < m) cnt += before(s_key[o], (unsigned)s_pos[o], ke, pe);
            }
            if (cnt) atomicAdd(&s_rank[e], cnt);
        }
        __syncthreads();
        if (tid < m) {
            const int r = s_rank[tid];
            const float lo_[4] = {lo4.x, lo4.y, lo4.z, lo4.w}, an_[4] = {an4.x, an4.y, an4.z, an4.w};
            const Box4 bx = decode_anchor(lo_, an_);
            s_key[r] = my_key; s_pos[r] = my_pos;
            s_box[r] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
            s_area[r] = (bx.x2 - bx.x1 + 1.0f) * (bx.y2 - bx.y1 + 1.0f);
        }
        __syncthreads();
    }
    if (one_wave) {
        NTRACE(6);
        // overlap matrix: row i, bits j = ovr(i, j) > thr, as ballots -- wave w takes rows w, w + 16, ...; lane = candidate j (and j + 64).
        // Symmetric in (i, j): max / min and one commutative sum per term, so a row read as a column is the same bits.
        {
            const float4 bj0 = lane < m ? s_box[lane] : make_float4(0.f, 0.f, 0.f, 0.f), bj1 = lane + 64 < m ? s_box[lane + 64] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float aj0 = lane < m ? s_area[lane] : 0.f, aj1 = lane + 64 < m ? s_area[lane + 64] : 0.f;
            for (int i = wv; i < m; i += kImgT / 64) {
                const float4 q = s_box[i]; const float ai = s_area[i];
                auto over = [&](const float4& tb, float ta) {
                    const float w = fmaxf(fminf(q.z, tb.z) - fmaxf(q.x, tb.x) + 1.0f, 0.0f);     // clamp(min=x1[i]) etc.
                    const float h = fmaxf(fminf(q.w, tb.w) - fmaxf(q.y, tb.y) + 1.0f, 0.0f);
                    const float inter = w * h;
                    return inter / (ta + ai - inter) > P.nms_thresh;
                };
                const unsigned long long lo = __ballot(lane < m && over(bj0, aj0));
                const unsigned long long hi = m > 64 ? __ballot(lane + 64 < m && over(bj1, aj1)) : 0ull;
                if (lane == 0) { s_rows[i][0] = lo; s_rows[i][1] = hi; }
            }
        }
        __syncthreads();
    } else {
        for (int i = m + tid; i < mp; i += kImgT) { s_key[i] = -INFINITY; s_pos[i] = 0x7fffffff; }
        __syncthreads();
        for (int k = 2; k <= mp; k <<= 1) {
            for (int lj = __ffs(k) - 2; lj >= 0; --lj) {
                const int j = 1 << lj;
                for (int t = tid; t < mp / 2; t += kImgT) {
                    const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                    const bool up = ((i & k) == 0);
                    const float ki = s_key[i], kl = s_key[l]; const int pi = s_pos[i], pl = s_pos[l];
                    const bool sw = up ? before(kl, (unsigned)pl, ki, (unsigned)pi) : before(ki, (unsigned)pi, kl, (unsigned)pl);
                    if (sw) { s_key[i] = kl; s_key[l] = ki; s_pos[i] = pl; s_pos[l] = pi; }
                }
                __syncthreads();
            }
        }
        if (tid < m) {
            const int a = cidx[s_pos[tid]];
            const Box4 bx = decode_anchor(P.loc + (o0 + a) * 4, P.anchors + (size_t)a * 4);
            s_box[tid] = make_float4(bx.x1, bx.y1, bx.x2, bx.y2);
            s_area[tid] = (bx.x2 - bx.x1 + 1.0f) * (bx.y2 - bx.y1 + 1.0f);
            s_alive[tid] = 1;
        }
        __syncthreads();
    }
    // ---- greedy loop (RetinaNet.py:143-169): stop WITHOUT keeping when one box is left.  ONE wave runs it out of LDS.
    if (wv == 0) {
        int K = 0;
        if (m <= 128) {
            // lane holds the rows of candidates lane and lane + 64 (bits 0-63 / 64-127); pivot = first alive bit; keep it;
            // alive &= ~(its row | itself).  The loop touches registers only (scalar masks, readlane for the pivot's row); the kept list
            // is written afterwards from the kept mask, in pivot order.
            const unsigned long long r0lo = s_rows[lane][0], r0hi = s_rows[lane][1], r1hi = s_rows[lane + 64][1];
            NTRACE(7);
            unsigned long long alo = m >= 64 ? ~0ull : ((1ull << m) - 1ull), ahi = m > 64 ? (m >= 128 ? ~0ull : ((1ull << (m - 64)) - 1ull)) : 0ull;
            unsigned long long keep_lo = 0ull, keep_hi = 0ull;
            auto rl64 = [&](unsigned long long v, int src) {
                return ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), src) << 32) |
                       (unsigned)__builtin_amdgcn_readlane((int)(unsigned)v, src);
            };
            // pivots of the low word first, then of the high word: two straight scalar loops (no pivot-side branch inside)
            while (alo != 0ull && __popcll(alo) + __popcll(ahi) > 1) {
                const int p = __ffsll((long long)alo) - 1;
                const unsigned long long bit = 1ull << p;
                const unsigned long long klo = rl64(r0lo, p), khi = rl64(r0hi, p);       // the pivot's row
                keep_lo |= bit;
                alo &= ~(klo | bit); ahi &= ~khi;
            }
            while (alo == 0ull && __popcll(ahi) > 1) {
                const int p = __ffsll((long long)ahi) - 1;
                const unsigned long long bit = 1ull << p;
                const unsigned long long khi = rl64(r1hi, p);                             // its low half only reaches candidates already gone
                keep_hi |= bit;
                ahi &= ~(khi | bit);
            }
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int nlo = __popcll(keep_lo);
            if ((keep_lo >> lane) & 1ull) s_keep[__popcll(keep_lo & lt)] = s_pos[lane];
            if ((keep_hi >> lane) & 1ull) s_keep[nlo + __popcll(keep_hi & lt)] = s_pos[lane + 64];
            K = nlo + __popcll(keep_hi);
        } else {
            int p = 0, alive = m;
            while (alive > 1) {
                const float4 tb = s_box[p]; const float ta = s_area[p];
                int nxt = INT_MAX, killed = 0;
                for (int j = lane; j < m; j += 64) {
                    if (j == p || !s_alive[j]) continue;
                    const float4 q = s_box[j];
                    const float w = fmaxf(fminf(q.z, tb.z) - fmaxf(q.x, tb.x) + 1.0f, 0.0f);     // clamp(min=x1[i]) etc.
                    const float h = fmaxf(fminf(q.w, tb.w) - fmaxf(q.y, tb.y) + 1.0f, 0.0f);
                    const float inter = w * h;
                    const float ovr = inter / (ta + s_area[j] - inter);
                    if (ovr <= P.nms_thresh) nxt = min(nxt, j); else { s_alive[j] = 0; ++killed; }
                }
                for (int o = 32; o > 0; o >>= 1) { nxt = min(nxt, __shfl_xor(nxt, o, 64)); killed += __shfl_xor(killed, o, 64); }
                if (lane == (p & 63)) { s_alive[p] = 0; s_keep[K] = s_pos[p]; }
                ++K;
                alive -= killed + 1;
                p = nxt == INT_MAX ? 0 : nxt;
            }
        }
        if (lane == 0) s_p = K;
    }
    NTRACE(8);
    __syncthreads();
    const int K = s_p;
    // ---- output rows (RetinaNet.py:170-176): boxes / labels indexed by the FILTERED position in the UNFILTERED arrays
    if (tid < K) {
        const int k = s_keep[tid];
        const Box4 bx = decode_anchor(P.loc + (o0 + k) * 4, P.anchors + (size_t)k * 4);
        float* row = out + (size_t)tid * 7;
        row[0] = bx.x1; row[1] = bx.y1; row[2] = bx.x2; row[3] = bx.y2; row[4] = 0.0f;
        row[5] = ckey[k]; row[6] = (float)P.label[o0 + k];
    }
    if (tid == 0) { P.out_counts[b * 2] = K; P.out_counts[b * 2 + 1] = n; }
    NTRACE(9);
}

#ifdef ODPL_NMS_TRACE
extern "C" int odpl_greedy_nms_trace(unsigned long long* buf) { return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nms_trace), &buf, sizeof(buf)); }
#endif

struct GPlan { size_t score, label, ckey, cidx, tile_cnt, ncand, total; int ntiles; };
bool make_gplan(const odpl_greedy_nms_desc_t* d, GPlan* p) {
    if (!d || d->B < 1 || d->An < 1 || d->An >= (1 << 24) || d->C < 1 || d->topk < 1 || d->topk > kMaxTopk) return false;
    if (!d->loc || !d->cls || !d->anchors) return false;
    if ((size_t)kRows * (d->C + 4) * 4 > 60 * 1024) return false;
    const size_t BN = (size_t)d->B * d->An;
    size_t off = 0;
    p->score = off; off += align_up(BN * 4, 256);
    p->label = off; off += align_up(BN * 4, 256);
    p->ckey = off;  off += align_up(BN * 4, 256);
    p->cidx = off;  off += align_up(BN * 4, 256);
    p->ntiles = (d->An + kRows - 1) / kRows;
    p->tile_cnt = off; off += align_up((size_t)d->B * p->ntiles * 4, 256);
    p->ncand = off; off += align_up((size_t)d->B * 4, 256);
    p->total = off;
    return true;
}
}  // namespace

extern "C" size_t odpl_greedy_nms_workspace_bytes(const odpl_greedy_nms_desc_t* d) {
    GPlan p;
    return make_gplan(d, &p) ? p.total : 0;
}

extern "C" int odpl_greedy_nms(const odpl_greedy_nms_desc_t* d, void* workspace, size_t workspace_bytes,
                               float* out_dets, int32_t* out_counts, void* stream_) {
    GPlan pl;
    if (!make_gplan(d, &pl) || !workspace || workspace_bytes < pl.total || !out_dets || !out_counts) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    GParams P;
    P.loc = d->loc; P.cls = d->cls; P.anchors = d->anchors; P.B = d->B; P.An = d->An; P.C = d->C; P.topk = d->topk;
    P.nms_thresh = d->nms_thresh; P.class_thresh = d->class_thresh;
    P.score = (float*)(ws + pl.score); P.label = (int*)(ws + pl.label); P.ckey = (float*)(ws + pl.ckey); P.cidx = (int*)(ws + pl.cidx);
    P.tile_cnt = (int*)(ws + pl.tile_cnt); P.ntiles = pl.ntiles; P.ncand = (int*)(ws + pl.ncand);
    P.magic_c = fastdiv16_magic((unsigned)d->C); P.magic_c4 = fastdiv16_magic((unsigned)(d->C >> 2));
    P.out = out_dets; P.out_counts = out_counts;
    prof_begin(ODPL_PROF_GREEDY_NMS_PREP, stream);
    if (d->C == 80) hipLaunchKernelGGL(greedy_nms_prep_ct_kernel<80>, dim3((d->An + kRows - 1) / kRows, d->B), dim3(256), (size_t)kRows * 80 * 4, stream, P);
    else hipLaunchKernelGGL(greedy_nms_prep_kernel, dim3((d->An + kRows - 1) / kRows, d->B), dim3(256), (size_t)kRows * (d->C + 4) * 4, stream, P);
    prof_end(ODPL_PROF_GREEDY_NMS_PREP, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_GREEDY_NMS_IMAGE, stream);
    hipLaunchKernelGGL(greedy_nms_scan_kernel, dim3(d->B), dim3(kImgT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(greedy_nms_scatter_kernel, dim3((pl.ntiles + (kScT / 64) * kScTU - 1) / ((kScT / 64) * kScTU), d->B), dim3(kScT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(greedy_nms_image_kernel, dim3(d->B), dim3(kImgT), 0, stream, P);
    prof_end(ODPL_PROF_GREEDY_NMS_IMAGE, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
