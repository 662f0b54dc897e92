// YOLO v2/v3/v4 region loss for MI355X (gfx950): target assignment + the seven loss terms +
// d loss / d head, replacing losses.py:475-905 and accuracy.py:305-380 of the reference.
//
// Data layout in HBM.  The head tensor is consumed exactly as the producing convolution wrote it, in either memory
// format of a [B, A*(5+C), G, G] tensor, and the gradient goes back in the same format:
//   planar      (NCHW)           element (b,a,c,s) at ((b*A + a)*(5+C) + c)*G2 + s      cell index (b*A + a)*G2 + s
//   interleaved (channels-last)  element (b,a,c,s) at ((b*G2 + s)*A + a)*(5+C) + c      cell index (b*G2 + s)*A + a
// The reference's view().permute(0,1,3,4,2).contiguous() copy (losses.py:517-522) is never made.
//
// What the loss touches.  Given the assignment, every term is local to a cell, and all terms but the confidence BCE live
// on the nObj <= nT "obj" cells only (losses.py:557-564).  The gradient is therefore STRUCTURED: one dense value per cell
// (its confidence logit) + 4+C values on each obj cell + zeros.  forward produces it in that compact form:
//   assign   (1 workgroup / scale)   targets -> int32 cell-state map {-1 noobj, 0 ignored, t+1 obj owner}, per-winner
//                                    records (tx,ty,tw,th, class bitset, raster rank), exact integer nObj / nNoobj
//   loss     (one launch)            "conf" blocks: confidence logit + state of every cell -> BCE(obj / noobj x100) partial
//                                    sums, gconf[cell] = d loss/d logit, state reset; "obj" blocks (1 / target / scale):
//                                    the 5+C logits of an obj cell -> x,y,w,h,cls terms, gobj[t][5+C], "Size" metric
//   finalize (1 workgroup)           fixed-order reduction -> the reference's 7 scalars per scale and total
// [r4] Up to kFusedMaxT targets (every training batch of the BASELINE configs) assign + loss are ONE launch with no state map and no
// global atomics (yolo_fused_kernel): the assignment is a function of the <= 512 target rows alone, so every workgroup derives
// what it needs of it from the rows themselves in LDS -- its scale's guards, winners (last row on a cell wins), distinct-cell counts
// nObj / nNoobj, and the state of ITS OWN cells -- while its logits are already in flight.  Same arithmetic, same partial sums, same
// bits as the two launches; one dependent launch (and its 3-4 dependent trips to memory) less: 6.8 + 7.0 us -> one launch.
// backward consumes the compact gradient:
//   expand + patch                   the dense tensor autograd hands to a generic producer: zeros streamed with 16-B
//                                    non-temporal stores and gconf in the confidence slots (HBM-write bound; the kernel the
//                                    generic path's roofline figure is quoted on), then gobj scattered onto the obj cells;
//                                    all scaled by the upstream scalar read on the device.
//   (or odpl_headgrad.hip)           when the producer is a 1x1 convolution the dense tensor is never materialised: the
//                                    convolution's backward is computed from the compact form directly.
// No atomics on floats anywhere: results are bitwise reproducible run to run.
#include <hip/hip_bf16.h>
#include "odpl_common.hpp"
#include "odpl_yolo_dev.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

constexpr int kAssignThreads = 256;
constexpr int kAssignRankCap = 4096;    // winners' cells staged in LDS for the raster-rank pass
constexpr int kConfThreads = 256;
constexpr int kObjThreads = 128;        // obj blocks use the first 128 threads of a loss-kernel workgroup
constexpr int kExpandThreads = 256;

// tunables (odpl_tune_set): defaults chosen on MI355X, see DESIGN.md "Kernels"
int g_fill_f4_per_block = 2048;         // float4 stores per expand block (x16 B)
unsigned long long* g_stamp = nullptr;
int g_fill_nt = 1;                      // 1: non-temporal stores for the zero stream (in-step: 11.4 vs 12.1 us)
constexpr int kFusedMaxT = 512;         // targets per call up to which assignment + loss are one launch (LDS tables, an O(nT) scan per row)
int g_fused_max_t = kFusedMaxT;         // odpl_tune_set(ODPL_TUNE_YOLO_FUSED_MAX_TARGETS): 0 = always assign launch + loss launch

struct Scale {
    const float* head;
    float* grad;         // expand destination
    int* state;          // [cells]  (persistent workspace; kept at -1 between calls: the conf blocks reset what they read)
    float* rec;          // [nT*4]   tx,ty,tw,th of target t (meaningful for winners)
    int* tcell;          // [nT]     linear cell (layout order) if target t owns its cell, else -1
    int* trank;          // [nT]     raster rank of the cell among obj cells (winners)
    unsigned* bits;      // [nT*W]   class bitset of the winner's cell (multi-hot on shared cells)
    int* counts;         // [8]
    float* conf_part;    // [conf_nblk*2]  (sum obj BCE, sum noobj BCE)
    float* obj_part;     // [nT*8]   lx, ly, lw, lh, lcls, size, -, -
    float* gconf;        // [cells]  compact gradient of the confidence logits (NULL: forward only)
    float* gobj;         // [nT*CH]  compact gradient of each winner's cell (slot 4 unused)
    int G, G2, A, layout, cells;
    int conf_blk0, conf_blk_per_row, conf_nblk, conf_vec;
    int obj_blk0;
    int exp_blk0, exp_nblk, rows_per_exp_blk;
    unsigned long long inv_ch;           // ceil(2^40 / CH): e / CH for e < 2^31 by one 64-bit multiply
    float aw[ODPL_MAX_ANCHORS], ah[ODPL_MAX_ANCHORS];
};

struct Params {
    Scale s[ODPL_MAX_SCALES];
    const float* target;
    const float* upstream;       // device scalar d L / d loss, or NULL (= 1)
    int nscales, B, C, CH, nT, W;
    int coord_sl1, cls_mode, average, fill_nt;
    float ignore_thres, obj_scale, noobj_scale, inv_S;
    float* out7;
    float* out_scales;
    int* out_counts;
    unsigned long long* stamp;   // diagnostic only (odpl_debug_set_stamp_buffer): [2*wg] start, [2*wg+1] end, 100 MHz ticks
};

__device__ __forceinline__ int cell_index(const Scale& S, int b, int a, int j, int i) {
    return S.layout == ODPL_LAYOUT_PLANAR ? ((b * S.A + a) * S.G + j) * S.G + i : ((b * S.G + j) * S.G + i) * S.A + a;
}

__global__ __launch_bounds__(kAssignThreads) void yolo_assign_kernel(const Params P) {
    const Scale& S = P.s[blockIdx.x];
    const int tid = threadIdx.x, nT = P.nT, G = S.G, A = S.A, B = P.B, C = P.C, W = P.W;
    __shared__ int sh_i[2 * (kAssignThreads / 64)];
    __shared__ int s_cell[kAssignRankCap];
    __shared__ float s_aw[ODPL_MAX_ANCHORS], s_ah[ODPL_MAX_ANCHORS];
    __shared__ int s_flags;
    if (tid < ODPL_MAX_ANCHORS) { s_aw[tid] = S.aw[tid]; s_ah[tid] = S.ah[tid]; }
    if (tid == 0) s_flags = 0;
    __syncthreads();
    // target `tid` stays decoded in registers (the common nT <= 256 case never decodes twice)
    TDec d0;
    if (tid < nT) d0 = decode_target(P.target + (size_t)tid * 6, G, A, s_aw, s_ah, P.ignore_thres);
    auto dec = [&](int t) { return t == tid ? d0 : decode_target(P.target + (size_t)t * 6, G, A, s_aw, s_ah, P.ignore_thres); };

    int flags = 0;
    for (int t = tid; t < nT; t += kAssignThreads) {
        const TDec d = dec(t);
        flags |= ((d.b >= B) | (d.best >= A) | (d.gj >= G) | (d.gi >= G)) ? 1 : 0;      // accuracy.py:340-343
        flags |= (d.lab >= C) ? 2 : 0;                                                   // accuracy.py:365
        flags |= ((d.b < -B) | (d.gj < -G) | (d.gi < -G) | (d.lab < -C)) ? 4 : 0;
        S.tcell[t] = -1; S.trank[t] = 0;
        for (int w = 0; w < W; ++w) S.bits[(size_t)t * W + w] = 0u;
        if (t < kAssignRankCap) s_cell[t] = -1;
    }
    // one barrier publishes all three flags
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    flags = s_flags;
    const int g1 = flags & 1 ? 1 : 0, g2 = (flags & 3) ? 1 : 0, st = flags & 4 ? 1 : 0;
    __syncthreads();                                                      // everyone has read s_flags before bit 3 may be set below

    int left = 0, became = 0;
    for (int t = tid; t < nT; t += kAssignThreads) {
        const TDec d = dec(t);
        long long b = d.b, gj = d.gj, gi = d.gi;
        const bool upper_ok = (d.b < B) & (d.gj < G) & (d.gi < G);       // per-target guard, accuracy.py:354-356
        const bool ok = wrap_idx(b, B) & wrap_idx(gj, G) & wrap_idx(gi, G);
        if (!ok) continue;
        if (!g1) {                                                        // accuracy.py:344-346
            const int cell = cell_index(S, (int)b, d.best, (int)gj, (int)gi);
            const int old = atomicMax(&S.state[cell], t + 1);
            left += (old == -1); became += (old < 1);
        }
        if (upper_ok) {                                                   // accuracy.py:349-358
            for (int a = 0; a < A; ++a) if ((d.over >> a) & 1u) {
                const int cell = cell_index(S, (int)b, a, (int)gj, (int)gi);
                const int old = atomicMax(&S.state[cell], 0);
                left += (old == -1);
            }
        }
    }
    __syncthreads();
    if (!g1) {
        for (int t = tid; t < nT; t += kAssignThreads) {
            const TDec d = dec(t);
            long long b = d.b, gj = d.gj, gi = d.gi, lab = d.lab;
            if (!(wrap_idx(b, B) & wrap_idx(gj, G) & wrap_idx(gi, G))) continue;
            const int cell = cell_index(S, (int)b, d.best, (int)gj, (int)gi);
            const int w = __hip_atomic_load(&S.state[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - 1;
            if ((unsigned)w >= (unsigned)nT) {                            // impossible on a clean map (the owner is a row of THIS call):
                atomicOr(&s_flags, 8);                                    // the STATE_CLEAN promise was broken -- flag it, index nothing
                continue;
            }
            if (w == t) {                                                 // last writer in row order wins
                S.tcell[t] = cell;
                if (t < kAssignRankCap) s_cell[t] = cell;
                float* r = S.rec + (size_t)t * 4;
                if (!g2) {                                                // accuracy.py:368-372
                    r[0] = d.gx - floorf(d.gx);
                    r[1] = d.gy - floorf(d.gy);
                    r[2] = logf(d.gw / s_aw[d.best] + 1e-16f);
                    r[3] = logf(d.gh / s_ah[d.best] + 1e-16f);
                } else { r[0] = r[1] = r[2] = r[3] = 0.f; }
            }
            if (!g2 && wrap_idx(lab, C))                                  // accuracy.py:374, multi-hot
                atomicOr(&S.bits[(size_t)w * W + (int)(lab >> 5)], 1u << (int)(lab & 31));
        }
    }
    __syncthreads();
    // raster rank of every obj cell (order of pred_boxes[..., 2:4][obj_mask], losses.py:568: (b, a, j, i) ascending).  The
    // interleaved cell index orders (b, j, i, a), so ranks compare the planar key; s_cell is rewritten with keys first.
    auto raster_key = [&](int c) {
        if (c < 0 || S.layout == ODPL_LAYOUT_PLANAR) return c;
        const int a = c % A, bs = c / A, b = bs / S.G2, s = bs - b * S.G2;
        return (b * A + a) * S.G2 + s;
    };
    if (S.layout != ODPL_LAYOUT_PLANAR) {
        for (int t = tid; t < nT && t < kAssignRankCap; t += kAssignThreads) s_cell[t] = raster_key(s_cell[t]);
        __syncthreads();
    }
    for (int t = tid; t < nT; t += kAssignThreads) {
        const int kc = t < kAssignRankCap ? s_cell[t] : raster_key(S.tcell[t]);
        if (kc < 0) continue;
        int r = 0;
        const int nl = nT < kAssignRankCap ? nT : kAssignRankCap;
        for (int u = 0; u < nl; ++u) { const int ku = s_cell[u]; r += (ku >= 0) & (ku < kc); }
        for (int u = kAssignRankCap; u < nT; ++u) { const int ku = raster_key(S.tcell[u]); r += (ku >= 0) & (ku < kc); }
        S.trank[t] = r;
    }
    // counts
    left = wave_sum_i(left); became = wave_sum_i(became);
    if ((tid & 63) == 0) { sh_i[(tid >> 6) * 2] = left; sh_i[(tid >> 6) * 2 + 1] = became; }
    __syncthreads();
    if (tid == 0) {
        int L = 0, Bc = 0;
        for (int w = 0; w < kAssignThreads / 64; ++w) { L += sh_i[w * 2]; Bc += sh_i[w * 2 + 1]; }
        S.counts[0] = Bc;                       // nObj
        S.counts[1] = B * A * S.G2 - L;         // nNoobj
        S.counts[2] = g1;
        S.counts[3] = g2;
        S.counts[4] = st | ((s_flags & 8) ? 2 : 0);     // status: bit0 index below -size (the reference raises), bit1 dirty state map
        S.counts[5] = S.counts[6] = S.counts[7] = 0;
    }
}

// ------------------------------------------------------------------------------------------------
// loss kernel: confidence term of every cell ("conf" blocks) + the other terms on the obj cells ("obj" blocks)

__device__ __forceinline__ void conf_cell(float z, int st, float w_obj, float w_noobj, float& g, float& lo, float& ln) {
    const float p = sigmoidf_(z);
    g = 0.f;
    if (st > 0) {                                   // obj: BCE(p, 1)   losses.py:561
        lo += -fmaxf(logf(p), -100.0f);
        g = bce_grad_logit(p, 1.0f, w_obj);
    } else if (st < 0) {                            // noobj: BCE(p, 0) losses.py:562
        ln += -fmaxf(log1pf(-p), -100.0f);
        g = bce_grad_logit(p, 0.0f, w_noobj);
    }                                               // st == 0: ignored (anchor IoU > thres)
}

// diagnostic stamps: first-start / last-end of the grid on the constant 100 MHz counter (never read by the kernel)
__device__ __forceinline__ void stamp_begin(const Params& P) {
    if (P.stamp && threadIdx.x == 0) P.stamp[2 * blockIdx.x] = (unsigned long long)wall_clock64();
}
__device__ __forceinline__ void stamp_end(const Params& P) {
    if (P.stamp) { __syncthreads(); if (threadIdx.x == 0) { __builtin_amdgcn_s_waitcnt(0); P.stamp[2 * blockIdx.x + 1] = (unsigned long long)wall_clock64(); } }
}

// The confidence logits of this workgroup's cells, requested before anything else the workgroup does (the fused launch decodes the
// target rows while they travel).  idx: float4 index within the plane (planar, vector), cell within the plane (planar, scalar) or
// global cell (interleaved); ok: the lane has cells.
struct ConfIn { float4 z; int idx; bool ok; };
__device__ __forceinline__ ConfIn conf_load(const Params& P, const Scale& S, int local) {
    const int tid = threadIdx.x, G2 = S.G2, CH = P.CH;
    ConfIn in; in.z = make_float4(0.f, 0.f, 0.f, 0.f); in.ok = false; in.idx = 0;
    if (S.layout == ODPL_LAYOUT_PLANAR) {
        const int plane = local / S.conf_blk_per_row, part = local - plane * S.conf_blk_per_row;
        const float* __restrict__ z = S.head + ((size_t)plane * CH + 4) * G2;
        in.idx = part * kConfThreads + tid;
        if (S.conf_vec) { in.ok = in.idx < G2 / 4; if (in.ok) in.z = reinterpret_cast<const float4*>(z)[in.idx]; }
        else { in.ok = in.idx < G2; if (in.ok) in.z.x = z[in.idx]; }
    } else {
        // interleaved: thread = cell; the logit sits CH floats from its neighbour's (one memory request per cell), the
        // state map and gconf are compact and coalesced
        in.idx = local * kConfThreads + tid;
        in.ok = in.idx < S.cells;
        if (in.ok) in.z.x = S.head[(size_t)in.idx * CH + 4];
    }
    return in;
}

// lstate != null (fused launch): the states of this workgroup's cells, in LDS, already final; counts come from the caller; the
// global state map is not touched.  lstate == null: the state map of the assign launch (read, then reset to -1).
__device__ __forceinline__ void conf_finish(const Params& P, const Scale& S, int local, const ConfIn& in, float* sh, const int* lstate,
                                            int nObj, int nNoobj) {
    const int tid = threadIdx.x, G2 = S.G2;
    const float w_obj = P.obj_scale * P.inv_S / (float)nObj;
    const float w_noobj = P.noobj_scale * P.inv_S / (float)nNoobj;
    float acc[2] = {0.f, 0.f};
    if (S.layout == ODPL_LAYOUT_PLANAR) {
        const int plane = local / S.conf_blk_per_row;
        float* __restrict__ g = S.gconf ? S.gconf + (size_t)plane * G2 : nullptr;
        int* __restrict__ st = lstate ? nullptr : S.state + (size_t)plane * G2;
        if (S.conf_vec) {
            if (in.ok) {
                const int i4 = in.idx;
                const float4 zv = in.z;
                const int4 sv = lstate ? reinterpret_cast<const int4*>(lstate)[tid] : reinterpret_cast<const int4*>(st)[i4];
                float4 gv;
                conf_cell(zv.x, sv.x, w_obj, w_noobj, gv.x, acc[0], acc[1]);
                conf_cell(zv.y, sv.y, w_obj, w_noobj, gv.y, acc[0], acc[1]);
                conf_cell(zv.z, sv.z, w_obj, w_noobj, gv.z, acc[0], acc[1]);
                conf_cell(zv.w, sv.w, w_obj, w_noobj, gv.w, acc[0], acc[1]);
                if (g) reinterpret_cast<float4*>(g)[i4] = gv;
                if (!lstate) reinterpret_cast<int4*>(st)[i4] = make_int4(-1, -1, -1, -1);      // leave the map clean for the next call
            }
        } else if (in.ok) {
            const int i = in.idx;
            float gv;
            conf_cell(in.z.x, lstate ? lstate[tid] : st[i], w_obj, w_noobj, gv, acc[0], acc[1]);
            if (g) g[i] = gv;
            if (!lstate) st[i] = -1;
        }
    } else if (in.ok) {
        const int c = in.idx;
        float gv;
        conf_cell(in.z.x, lstate ? lstate[tid] : S.state[c], w_obj, w_noobj, gv, acc[0], acc[1]);
        if (S.gconf) S.gconf[c] = gv;
        if (!lstate) S.state[c] = -1;
    }
    block_sum<2>(acc, sh);
    if (tid == 0) { S.conf_part[local * 2] = acc[0]; S.conf_part[local * 2 + 1] = acc[1]; }
}

// first cell (layout order) and number of cells of conf workgroup `local` -- the range its LDS state table covers
__device__ __forceinline__ void conf_range(const Scale& S, int local, int& c0, int& n) {
    if (S.layout == ODPL_LAYOUT_PLANAR) {
        const int plane = local / S.conf_blk_per_row, part = local - plane * S.conf_blk_per_row;
        const int per = S.conf_vec ? 4 * kConfThreads : kConfThreads;
        c0 = plane * S.G2 + part * per;
        n = S.G2 - part * per; if (n > per) n = per;
    } else {
        c0 = local * kConfThreads;
        n = S.cells - c0; if (n > kConfThreads) n = kConfThreads;
    }
}

// one workgroup per target that may own an obj cell; the first kObjThreads threads work, all join the barriers.
// cell: the target's obj cell (layout order) if it owns one, else < 0; r: tx, ty, tw, th; bits: class bitset of the cell; rk: raster
// rank of the cell among the obj cells -- from the assign launch's workspace, or from the fused launch's LDS
__device__ __forceinline__ void obj_block(const Params& P, const Scale& S, int t, int cell, const float* r, const unsigned* bits, int rk,
                                          int nObj, float* sh, float* sh_bc) {
    const int tid = threadIdx.x, CH = P.CH, C = P.C, G2 = S.G2;
    const bool act = tid < kObjThreads;
    float* part = S.obj_part + (size_t)t * 8;
    if (cell < 0 || cell >= S.cells) { if (tid < 8) part[tid] = 0.f; return; }       // uniform per block
    size_t base, cs;
    int a;
    if (S.layout == ODPL_LAYOUT_PLANAR) {
        const int plane = cell / G2, s = cell - plane * G2;
        a = plane % S.A; base = (size_t)plane * CH * G2 + s; cs = (size_t)G2;
    } else {
        a = cell % S.A; base = (size_t)cell * CH; cs = 1;
    }
    const float* __restrict__ zp = S.head + base;
    float* __restrict__ go = S.gobj ? S.gobj + (size_t)t * CH : nullptr;
    const float w = P.inv_S / (float)nObj;                 // mean over nObj, then the 1/len(output)
    const int sl1 = P.coord_sl1;

    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};        // lx, ly, lw, lh, lcls, size
    // CE over the sigmoid outputs needs the row's log-sum-exp first (nn.CrossEntropyLoss, prob targets)
    float lse = 0.f, tsum = 0.f;
    if (P.cls_mode == ODPL_CLS_CE) {
        float m = -INFINITY, ts = 0.f;
        if (act) for (int k = tid; k < C; k += kObjThreads) {
            m = fmaxf(m, sigmoidf_(zp[(size_t)(5 + k) * cs]));
            ts += (float)((bits[k >> 5] >> (k & 31)) & 1u);
        }
        m = wave_max(m);
        if ((tid & 63) == 0) sh[tid >> 6] = m;
        __syncthreads();
        m = sh[0]; for (int i = 1; i < kObjThreads / 64; ++i) m = fmaxf(m, sh[i]);
        __syncthreads();
        float e[2] = {0.f, ts};
        if (act) for (int k = tid; k < C; k += kObjThreads) e[0] += expf(sigmoidf_(zp[(size_t)(5 + k) * cs]) - m);
        block_sum<2>(e, sh);
        if (tid == 0) { sh_bc[0] = m + logf(e[0]); sh_bc[1] = e[1]; }
        __syncthreads();
        lse = sh_bc[0]; tsum = sh_bc[1];
    }
    if (act) for (int c = tid; c < CH; c += kObjThreads) {
        if (c == 4) { if (go) go[4] = 0.f; continue; }     // confidence: conf blocks (gconf)
        const float z = zp[(size_t)c * cs];
        float g;
        if (c < 2) {                                       // losses.py:557-558
            const float p = sigmoidf_(z), d = p - r[c];
            acc[c] += coord_elem(d, sl1);
            g = coord_deriv(d, sl1) * w * (1.0f - p) * p;
        } else if (c < 4) {                                // losses.py:559-560 (raw, no sigmoid)
            const float d = z - r[c];
            acc[c] += coord_elem(d, sl1);
            g = coord_deriv(d, sl1) * w;
        } else {                                           // losses.py:564
            const int k = c - 5;
            const float tk = (float)((bits[k >> 5] >> (k & 31)) & 1u);
            const float p = sigmoidf_(z);
            if (P.cls_mode == ODPL_CLS_CE) {
                const float lsm = p - lse;
                acc[4] += -tk * lsm;
                g = (expf(lsm) * tsum - tk) * w * (1.0f - p) * p;
            } else {
                acc[4] += bce_elem(p, tk);
                g = bce_grad_logit(p, tk, w / (float)C);
            }
        }
        if (go) go[c] = g;
    }
    // "Size" metric (losses.py:568-573): obj cells in raster order vs target COLUMNS 3:5 of row `rank`
    if (tid == 0) {
        const int k = nObj < P.nT ? nObj : P.nT;
        if (rk < k) {
            const float pw = expf(zp[(size_t)2 * cs]) * S.aw[a], ph = expf(zp[(size_t)3 * cs]) * S.ah[a];
            const float* yr = P.target + (size_t)rk * 6;
            const float d0 = sqrtf(fabsf(pw) + 1e-32f) - sqrtf(fabsf(yr[3]) + 1e-32f);
            const float d1 = sqrtf(fabsf(ph) + 1e-32f) - sqrtf(fabsf(yr[4]) + 1e-32f);
            acc[5] = coord_elem(d0, sl1) + coord_elem(d1, sl1);
        }
    }
    block_sum<6>(acc, sh);
    if (tid == 0) {
        for (int i = 0; i < 6; ++i) part[i] = acc[i];
        part[6] = part[7] = 0.f;
    }
}

__global__ __launch_bounds__(kConfThreads) void yolo_loss_kernel(const Params P) {
    __shared__ float sh[8 * (kConfThreads / 64)];
    __shared__ float sh_bc[2];
    const int bid = blockIdx.x;
    int si = 0;
    bool is_conf = true;
    for (int i = 0; i < P.nscales; ++i) {
        const Scale& T = P.s[i];
        if (bid >= T.conf_blk0 && bid < T.conf_blk0 + T.conf_nblk) { si = i; is_conf = true; }
        if (bid >= T.obj_blk0 && bid < T.obj_blk0 + P.nT) { si = i; is_conf = false; }
    }
    const Scale& S = P.s[si];
    if (is_conf) { const ConfIn in = conf_load(P, S, bid - S.conf_blk0); conf_finish(P, S, bid - S.conf_blk0, in, sh, nullptr, S.counts[0], S.counts[1]); }
    else {
        const int t = bid - S.obj_blk0;
        obj_block(P, S, t, S.tcell[t], S.rec + (size_t)t * 4, S.bits + (size_t)t * P.W, S.trank[t], S.counts[0], sh, sh_bc);
    }
}

// ------------------------------------------------------------------------------------------------
// [r4] assignment + loss in one launch (nT <= kFusedMaxT).  Grid as yolo_loss_kernel.  Every workgroup:
//   1. requests its logits (conf workgroups; they do not depend on the assignment),
//   2. decodes the target rows of ITS scale into LDS: position key (b, gj, gi) or -1, best anchor, ignore mask, validity -- the
//      per-row guards of accuracy.py:340-367 exactly as yolo_assign_kernel applies them,
//   3. scans the table once per own row: a row WINS its cell unless a later row claims the same (cell, best anchor) ("last row
//      in row order wins", as index_put does); its ignore touches are NEW unless an obj touch or an earlier row's ignore touch
//      sits on the same cell.  nObj = winners, nNoobj = cells - (winners + new ignore touches): the integers the state map's
//      atomicMax returns used to count,
//   4. conf workgroups: scatter the touches that fall into their own <= 1024 cells into an LDS state table, then the confidence
//      terms as ever; obj workgroups: the class bitset (rows sharing the winner's cell) and the raster rank from the same table,
//      then the x, y, w, h, class terms as ever.
// No global state map, no global atomics, nothing to keep clean between calls.
struct FusedTables {
    __attribute__((aligned(16))) int pos[kFusedMaxT];            // (b * G + gj) * G + gi after python's negative-index wrap, or -1 (the reference would skip / raise)
    unsigned meta[kFusedMaxT];      // bits 0-2 best anchor | 8-15 ignore mask (per-target guard passed) | 16 obj touch valid | 17 winner
    int lab[kFusedMaxT];            // wrapped label if it may set a class bit (guard 2 passed), else -1
};

__global__ __launch_bounds__(kConfThreads) void yolo_fused_kernel(const Params P) {
    __shared__ float sh[8 * (kConfThreads / 64)];
    __shared__ float sh_bc[2];
    __shared__ FusedTables T;
    __shared__ __attribute__((aligned(16))) int lstate[4 * kConfThreads];
    __shared__ int s_flags, s_cnt[2];
    // Which rows share their position (b, gj, gi) with another row?  Almost none do, and only those need the scan of step 3: every row
    // sets its position's bit in `seen`; a row that finds the bit already set marks the position in `twice`.  65 536 positions
    // (B * G^2: 43 264 at config 2's finest scale); larger maps scan every row.
    constexpr int kBmWords = 2048;
    __shared__ unsigned bm_seen[kBmWords], bm_twice[kBmWords];
    __shared__ unsigned s_bits[8];          // C <= 256 classes on this path (checked on the host)
    __shared__ int s_rank;
    __shared__ float s_rec[4];
    __shared__ float s_aw[ODPL_MAX_ANCHORS], s_ah[ODPL_MAX_ANCHORS];
    const int bid = blockIdx.x, tid = threadIdx.x, nT = P.nT, B = P.B, C = P.C;
    int si = 0;
    bool is_conf = true;
    for (int i = 0; i < P.nscales; ++i) {
        const Scale& Q = P.s[i];
        if (bid >= Q.conf_blk0 && bid < Q.conf_blk0 + Q.conf_nblk) { si = i; is_conf = true; }
        if (bid >= Q.obj_blk0 && bid < Q.obj_blk0 + P.nT) { si = i; is_conf = false; }
    }
    const Scale& S = P.s[si];
    const int G = S.G, A = S.A, G2 = S.G2;
    const int own = is_conf ? -1 : bid - S.obj_blk0;
    if (tid < ODPL_MAX_ANCHORS) { s_aw[tid] = S.aw[tid]; s_ah[tid] = S.ah[tid]; }
    if (tid == 0) { s_flags = 0; s_cnt[0] = s_cnt[1] = 0; s_rank = 0; }
    if (tid < 8) s_bits[tid] = 0u;
    int c0 = 0, cn = 0;
    ConfIn cin; cin.ok = false;
    if (is_conf) {
        cin = conf_load(P, S, bid - S.conf_blk0);       // 1. in flight while the rows are decoded
        conf_range(S, bid - S.conf_blk0, c0, cn);
        for (int i = tid; i < 4 * kConfThreads; i += kConfThreads) lstate[i] = -1;
    }
    for (int t = nT + tid; t < ((nT + 3) & ~3); t += kConfThreads) T.pos[t] = -2;      // pad the scan's last vector (matches nothing)
    const int npos = B * G2;
    const bool use_bm = npos <= 32 * kBmWords;
    if (use_bm) for (int i = tid; i < (npos + 31) >> 5; i += kConfThreads) { bm_seen[i] = 0u; bm_twice[i] = 0u; }
    __syncthreads();
    // 2. decode; flags as yolo_assign_kernel's first pass
    int flags = 0;
    for (int t = tid; t < nT; t += kConfThreads) {
        const TDec d = decode_target(P.target + (size_t)t * 6, G, A, s_aw, s_ah, P.ignore_thres);
        flags |= ((d.b >= B) | (d.best >= A) | (d.gj >= G) | (d.gi >= G)) ? 1 : 0;      // accuracy.py:340-343
        flags |= (d.lab >= C) ? 2 : 0;                                                   // accuracy.py:365
        flags |= ((d.b < -B) | (d.gj < -G) | (d.gi < -G) | (d.lab < -C)) ? 4 : 0;
        long long b = d.b, gj = d.gj, gi = d.gi, lab = d.lab;
        const bool upper_ok = (d.b < B) & (d.gj < G) & (d.gi < G);                       // accuracy.py:354-356
        const bool ok = wrap_idx(b, B) & wrap_idx(gj, G) & wrap_idx(gi, G);
        const int pp = ok ? (int)((b * G + gj) * G + gi) : -1;
        T.pos[t] = pp;
        if (use_bm && pp >= 0 && ((atomicOr(&bm_seen[pp >> 5], 1u << (pp & 31)) >> (pp & 31)) & 1u)) atomicOr(&bm_twice[pp >> 5], 1u << (pp & 31));
        T.meta[t] = (unsigned)d.best | ((ok && upper_ok) ? (d.over & 0xffu) << 8 : 0u) | (ok ? 1u << 16 : 0u);
        T.lab[t] = wrap_idx(lab, C) ? (int)lab : -1;
        if (t == own) {
            s_rec[0] = d.gx - floorf(d.gx); s_rec[1] = d.gy - floorf(d.gy);              // accuracy.py:368-372
            s_rec[2] = logf(d.gw / s_aw[d.best] + 1e-16f); s_rec[3] = logf(d.gh / s_ah[d.best] + 1e-16f);
        }
    }
    if (flags) atomicOr(&s_flags, flags);
    __syncthreads();
    flags = s_flags;
    const int g1 = flags & 1 ? 1 : 0, g2 = (flags & 3) ? 1 : 0, st = flags & 4 ? 1 : 0;
    // 3. one scan of the table per own row
    int n_win = 0, n_new = 0;
    for (int t = tid; t < nT; t += kConfThreads) {
        const int p = T.pos[t];
        unsigned m = T.meta[t];
        if (g1) m &= ~(1u << 16);                       // accuracy.py:344-346: one bad row switches every obj touch off
        const unsigned best = m & 7u, over = (m >> 8) & 0xffu;
        bool later_same = false;
        // anchors with an obj touch on this position (any row; the row itself first) / an ignore touch by an earlier row
        unsigned obj_here = ((m >> 16) & 1u) ? 1u << best : 0u, ign_before = 0u;
        auto other = [&](int u) {                       // another row on the same position: rare
            const unsigned mu = T.meta[u];
            const bool objv = !g1 && ((mu >> 16) & 1u);
            if (objv) { obj_here |= 1u << (mu & 7u); later_same |= (u > t) && (mu & 7u) == best; }
            if (u < t) ign_before |= (mu >> 8) & 0xffu;
        };
        if (p >= 0 && !(use_bm && !((bm_twice[p >> 5] >> (p & 31)) & 1u))) {
            // four position keys per LDS read, no dependent branch in the common case (a data-dependent `continue` per key cost one
            // LDS round trip per row: 5 us at 112 rows); positions are unique but for duplicates, which take the branch
            const int4* pos4 = reinterpret_cast<const int4*>(T.pos);
            const int n4 = (nT + 3) >> 2;
#pragma unroll 4
            for (int u4 = 0; u4 < n4; ++u4) {
                const int4 q = pos4[u4];
                const int u = u4 * 4;
                if (((q.x == p) & (u != t)) | ((q.y == p) & (u + 1 != t)) | ((q.z == p) & (u + 2 != t)) | ((q.w == p) & (u + 3 != t))) {
                    if (q.x == p && u != t) other(u);
                    if (q.y == p && u + 1 != t) other(u + 1);
                    if (q.z == p && u + 2 != t) other(u + 2);
                    if (q.w == p && u + 3 != t) other(u + 3);
                }
            }
        }
        const bool win = p >= 0 && ((m >> 16) & 1u) && !later_same;
        if (win) m |= 1u << 17;
        T.meta[t] = m;
        n_win += win;
        n_new += __popc(over & ~obj_here & ~ign_before);
    }
    n_win = wave_sum_i(n_win); n_new = wave_sum_i(n_new);
    if ((tid & 63) == 0 && (n_win | n_new)) { atomicAdd(&s_cnt[0], n_win); atomicAdd(&s_cnt[1], n_new); }     // integer adds: order-free
    __syncthreads();
    const int nObj = s_cnt[0], nNoobj = B * A * G2 - (s_cnt[0] + s_cnt[1]);
    if (is_conf) {
        if (bid == S.conf_blk0 && tid == 0) {           // one workgroup per scale publishes the counts (finalize, out_counts)
            S.counts[0] = nObj; S.counts[1] = nNoobj; S.counts[2] = g1; S.counts[3] = g2; S.counts[4] = st;
            S.counts[5] = S.counts[6] = S.counts[7] = 0;
        }
        // 4a. the touches that land in this workgroup's cells
        for (int t = tid; t < nT; t += kConfThreads) {
            const int p = T.pos[t];
            if (p < 0) continue;
            const unsigned m = T.meta[t];
            const int b = p / G2, s = p - b * G2;
            unsigned touch = (m >> 8) & 0xffu;          // ignore touches: state 0
            const unsigned objbit = ((m >> 16) & 1u) ? 1u << (m & 7u) : 0u;
            touch |= objbit;
            while (touch) {
                const int a = __ffs(touch) - 1; touch &= touch - 1;
                const int cell = S.layout == ODPL_LAYOUT_PLANAR ? (b * A + a) * G2 + s : (b * G2 + s) * A + a;
                if (cell >= c0 && cell < c0 + cn) atomicMax(&lstate[cell - c0], (objbit >> a) & 1u ? 1 : 0);
            }
        }
        __syncthreads();
        conf_finish(P, S, bid - S.conf_blk0, cin, sh, lstate, nObj, nNoobj);
        return;
    }
    // 4b. obj workgroup of row `own`
    const unsigned mo = T.meta[own];
    const bool win = (mo >> 17) & 1u;
    const int po = T.pos[own];
    int cell = -1, rkey_own = 0;
    if (win) {
        const int b = po / G2, s = po - b * G2, a = (int)(mo & 7u);
        cell = S.layout == ODPL_LAYOUT_PLANAR ? (b * A + a) * G2 + s : (b * G2 + s) * A + a;
        rkey_own = (b * A + a) * G2 + s;                // raster order of pred_boxes[obj_mask]: (b, a, j, i) ascending (losses.py:568)
        int below = 0;
        for (int u = tid; u < nT; u += kConfThreads) {
            const unsigned mu = T.meta[u];
            const int pu = T.pos[u];
            if (pu < 0) continue;
            if ((mu >> 17) & 1u) { const int bu = pu / G2, su = pu - bu * G2; below += ((bu * A + (int)(mu & 7u)) * G2 + su) < rkey_own; }
            // class bits: every row that sits on the winner's cell (valid obj touch) sets its label (accuracy.py:374, multi-hot)
            if (!g2 && pu == po && ((mu >> 16) & 1u) && (mu & 7u) == (mo & 7u) && T.lab[u] >= 0) atomicOr(&s_bits[T.lab[u] >> 5], 1u << (T.lab[u] & 31));
        }
        below = wave_sum_i(below);
        if ((tid & 63) == 0 && below) atomicAdd(&s_rank, below);
    }
    if (tid == 0) S.tcell[own] = cell;
    if (win && g2 && tid < 4) s_rec[tid] = 0.f;         // guard 2: tx..th stay zero (accuracy.py:365-372 skipped)
    __syncthreads();
    obj_block(P, S, own, cell, s_rec, s_bits, s_rank, nObj, sh, sh_bc);
}

// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64 * ODPL_MAX_SCALES) void yolo_finalize_kernel(const Params P) {
    // wave w reduces scale w (fixed lane-strided order + fixed shuffle tree => reproducible);
    // then lanes 0..6 of wave 0 add the scales in order (losses.py:606-612)
    __shared__ float res[ODPL_MAX_SCALES][8];
    const int tid = threadIdx.x, lane = tid & 63, si = tid >> 6;
    if (si < P.nscales) {
        const Scale& S = P.s[si];
        const int nObj = S.counts[0], nNoobj = S.counts[1];
        double a[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // lx ly lw lh lcls size conf_obj conf_noobj
        for (int t = lane; t < P.nT; t += 64) {
            const float4* p = reinterpret_cast<const float4*>(S.obj_part + (size_t)t * 8);
            const float4 u = p[0], v = p[1];
            a[0] += (double)u.x; a[1] += (double)u.y; a[2] += (double)u.z; a[3] += (double)u.w;
            a[4] += (double)v.x; a[5] += (double)v.y;
        }
        for (int b = lane; b < S.conf_nblk; b += 64) {
            const float2 c = reinterpret_cast<const float2*>(S.conf_part)[b];
            a[6] += (double)c.x; a[7] += (double)c.y;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            double v = a[i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            a[i] = v;
        }
        if (lane == 0) {
            float s[8];
            for (int i = 0; i < 8; ++i) s[i] = (float)a[i];
            const float fo = (float)nObj;
            const float lx = s[0] / fo, ly = s[1] / fo, lw = s[2] / fo, lh = s[3] / fo;
            const float lcls = P.cls_mode == ODPL_CLS_CE ? s[4] / fo : s[4] / (fo * (float)P.C);
            const float lobj = s[6] / fo, lnoobj = s[7] / (float)nNoobj;
            const float lconf = P.obj_scale * lobj + P.noobj_scale * lnoobj;          // losses.py:563
            const int k = nObj < P.nT ? nObj : P.nT;
            const float size = s[5] / (float)(2 * k);
            float r[7];
            r[0] = lx + ly + lw + lh + lconf + lcls;                                  // losses.py:566
            r[1] = lx + ly; r[2] = size; r[3] = lconf; r[4] = lcls; r[5] = lobj; r[6] = lnoobj;
            for (int i = 0; i < 7; ++i) { res[si][i] = r[i]; if (P.out_scales) P.out_scales[si * 7 + i] = r[i]; }
            if (P.out_counts) for (int i = 0; i < 8; ++i) P.out_counts[si * 8 + i] = S.counts[i];
        }
    }
    __syncthreads();
    if (tid < 7) {
        float acc = 0.f;
        for (int i = 0; i < P.nscales; ++i) acc += res[i][tid];                       // losses.py:606-612
        P.out7[tid] = P.average ? acc / (float)P.nscales : acc;                       // losses.py:616-623
    }
}

// ------------------------------------------------------------------------------------------------
// expand: compact gradient -> the dense d loss / d head tensor in the head's own memory format

template <bool NT>
__device__ __forceinline__ void st4(v4f* p, const v4f& v) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; }

template <bool NT>
__device__ __forceinline__ void expand_block(const Params& P, const Scale& S, int local) {
    const int tid = threadIdx.x, G2 = S.G2, CH = P.CH;
    const float f = P.upstream ? *P.upstream : 1.0f;
    if (S.layout == ODPL_LAYOUT_PLANAR) {
        // rows [r0, r1) of this scale; row q = plane*CH + c holds G2 floats: zeros, or gconf for c == 4
        const int R = P.B * S.A * CH;
        const int r0 = local * S.rows_per_exp_blk;
        int r1 = r0 + S.rows_per_exp_blk; if (r1 > R) r1 = R;
        int cur = r0;
        while (cur < r1) {
            const int c = cur % CH;
            const int next_conf = cur + ((4 - c) % CH + CH) % CH;     // first row >= cur with c == 4
            const int seg_end = next_conf < r1 ? next_conf : r1;
            if (seg_end > cur) fill_zero<NT, kExpandThreads>(S.grad, (size_t)cur * G2, (size_t)seg_end * G2);
            if (next_conf < r1) {
                const float* __restrict__ src = S.gconf + (size_t)(next_conf / CH) * G2;
                float* __restrict__ dst = S.grad + (size_t)next_conf * G2;
                if (S.conf_vec) {
                    for (int i4 = tid; i4 < G2 / 4; i4 += kExpandThreads) {
                        float4 v = reinterpret_cast<const float4*>(src)[i4];
                        v.x *= f; v.y *= f; v.z *= f; v.w *= f;
                        reinterpret_cast<float4*>(dst)[i4] = v;
                    }
                } else {
                    for (int i = tid; i < G2; i += kExpandThreads) dst[i] = src[i] * f;
                }
            }
            cur = seg_end + (next_conf < r1 ? 1 : 0);
        }
    } else {
        // interleaved: a linear chunk of the [cells*CH] array; element e = cell*CH + c, confidence slots at c == 4
        const size_t total = (size_t)S.cells * CH, totalq = total / 4;
        const size_t chunk = (size_t)S.rows_per_exp_blk;              // float4 per block
        const size_t q0 = (size_t)local * chunk;
        const bool last = q0 + chunk >= totalq;
        const size_t q1 = last ? totalq : q0 + chunk;
        v4f* g4 = reinterpret_cast<v4f*>(S.grad);
        for (size_t q = q0 + tid; q < q1; q += kExpandThreads) {
            const unsigned e = (unsigned)(q * 4);
            const unsigned c = (unsigned)(((unsigned long long)e * S.inv_ch) >> 40);     // cell of the quad's first element
            const unsigned r = e - c * (unsigned)CH;                                      // its channel
            v4f o = {0.f, 0.f, 0.f, 0.f};
            if (r >= 1u && r <= 4u) o[4 - r] = S.gconf[c] * f;        // the quad holds cell c's confidence slot (a quad that spills
                                                                      // into cell c+1 reaches at most its channel 2)
            st4<NT>(g4 + q, o);
        }
        if (last) for (size_t e = totalq * 4 + tid; e < total; e += kExpandThreads) {
            const size_t c = e / CH;
            S.grad[e] = (e - c * CH == 4) ? S.gconf[c] * f : 0.f;
        }
    }
}

__global__ __launch_bounds__(kExpandThreads) void yolo_expand_kernel(const Params P) {
    const int bid = blockIdx.x;
    stamp_begin(P);
    int si = 0;
    for (int i = 0; i < P.nscales; ++i) if (bid >= P.s[i].exp_blk0 && bid < P.s[i].exp_blk0 + P.s[i].exp_nblk) si = i;
    const Scale& S = P.s[si];
    if (P.fill_nt) expand_block<true>(P, S, bid - S.exp_blk0); else expand_block<false>(P, S, bid - S.exp_blk0);
    stamp_end(P);
}

// the 4+C non-confidence gradients of each obj cell, written over the zeros of the expand launch (stream order)
__global__ __launch_bounds__(kObjThreads) void yolo_patch_kernel(const Params P) {
    const Scale& S = P.s[blockIdx.y];
    if (!S.grad) return;
    const int t = blockIdx.x, CH = P.CH;
    const int cell = S.tcell[t];
    if (cell < 0 || cell >= S.cells) return;
    const float f = P.upstream ? *P.upstream : 1.0f;
    const float* go = S.gobj + (size_t)t * CH;
    size_t base, cs;
    if (S.layout == ODPL_LAYOUT_PLANAR) { const int plane = cell / S.G2, s = cell - plane * S.G2; base = (size_t)plane * CH * S.G2 + s; cs = (size_t)S.G2; }
    else { base = (size_t)cell * CH; cs = 1; }
    for (int c = threadIdx.x; c < CH; c += kObjThreads) if (c != 4) S.grad[base + (size_t)c * cs] = go[c] * f;
}

__global__ void scale_if_not_one_kernel(float* __restrict__ g, size_t n, const float* __restrict__ s) {
    const float f = *s;
    if (f == 1.0f) return;
    const size_t n4 = n / 4, stride = (size_t)gridDim.x * blockDim.x;
    float4* g4 = reinterpret_cast<float4*>(g);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = g4[i]; v.x *= f; v.y *= f; v.z *= f; v.w *= f; g4[i] = v;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] *= f;
}

__global__ void scale_if_not_one_bf16_kernel(__hip_bfloat16* __restrict__ g, size_t n, const float* __restrict__ s) {
    const float f = *s;
    if (f == 1.0f) return;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) g[i] = __float2bfloat16(__bfloat162float(g[i]) * f);
}

// ------------------------------------------------------------------------------------------------
inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Plan {
    // persistent + per-call scratch workspace
    size_t state_off[ODPL_MAX_SCALES], rec_off[ODPL_MAX_SCALES], trank_off[ODPL_MAX_SCALES], bits_off[ODPL_MAX_SCALES],
        counts_off[ODPL_MAX_SCALES], confp_off[ODPL_MAX_SCALES], objp_off[ODPL_MAX_SCALES], tcell_ws_off[ODPL_MAX_SCALES];
    size_t state_bytes_total, ws_total;
    // compact gradient buffer (survives until backward)
    size_t gconf_off[ODPL_MAX_SCALES], gobj_off[ODPL_MAX_SCALES], tcell_off[ODPL_MAX_SCALES], compact_total;
    int layout[ODPL_MAX_SCALES], cells[ODPL_MAX_SCALES];
    int conf_blk_per_row[ODPL_MAX_SCALES], conf_nblk[ODPL_MAX_SCALES], conf_vec[ODPL_MAX_SCALES];
    int exp_nblk[ODPL_MAX_SCALES], rows_per_exp_blk[ODPL_MAX_SCALES];
    int W;
};

// memory format of one scale from its element strides (b, c, y, x); all zero = NCHW contiguous
int classify_layout(const odpl_yolo_scale_t& s, int B, int CH) {
    const long long G = s.G, ACH = (long long)s.A * CH;
    const int64_t* t = s.stride;
    if (t[0] == 0 && t[1] == 0 && t[2] == 0 && t[3] == 0) return ODPL_LAYOUT_PLANAR;
    const bool b_ok = B == 1 || t[0] == ACH * G * G;
    if (b_ok && t[1] == G * G && (G == 1 || (t[2] == G && t[3] == 1))) return ODPL_LAYOUT_PLANAR;
    if (b_ok && t[1] == 1 && (G == 1 || (t[2] == G * ACH && t[3] == ACH))) return ODPL_LAYOUT_INTERLEAVED;
    return -1;
}

bool make_plan(const odpl_yolo_loss_desc_t* d, Plan* p) {
    if (!d || d->nscales < 1 || d->nscales > ODPL_MAX_SCALES || d->B < 1 || d->C < 1 || d->nT < 0) return false;
    const int CH = 5 + d->C;
    p->W = (d->C + 31) / 32;
    size_t off = 0;
    // all state maps first, contiguous: one fill(0xFF) resets them
    for (int i = 0; i < d->nscales; ++i) {
        const odpl_yolo_scale_t& s = d->scale[i];
        if (s.G < 1 || s.A < 1 || s.A > ODPL_MAX_ANCHORS || !s.head) return false;
        if ((long long)d->B * s.A * CH * s.G * s.G > 0x7fffffffLL) return false;
        p->layout[i] = classify_layout(s, d->B, CH);
        if (p->layout[i] < 0) return false;
        p->cells[i] = d->B * s.A * s.G * s.G;
        p->state_off[i] = off;
        off += align_up((size_t)p->cells[i] * sizeof(int), 256);
    }
    p->state_bytes_total = off;
    const size_t nT = (size_t)(d->nT > 0 ? d->nT : 1);
    size_t coff = 0;
    for (int i = 0; i < d->nscales; ++i) {
        const odpl_yolo_scale_t& s = d->scale[i];
        const int G2 = s.G * s.G;
        if (p->layout[i] == ODPL_LAYOUT_PLANAR) {
            p->conf_vec[i] = (G2 % 4 == 0) ? 1 : 0;
            const int units = p->conf_vec[i] ? G2 / 4 : G2;
            p->conf_blk_per_row[i] = (units + kConfThreads - 1) / kConfThreads;
            p->conf_nblk[i] = d->B * s.A * p->conf_blk_per_row[i];
            int rows = (g_fill_f4_per_block * 4 + G2 - 1) / G2; if (rows < 1) rows = 1;
            p->rows_per_exp_blk[i] = rows;
            const int R = d->B * s.A * CH;
            p->exp_nblk[i] = (R + rows - 1) / rows;
        } else {
            p->conf_vec[i] = 0;
            p->conf_blk_per_row[i] = 1;
            p->conf_nblk[i] = (p->cells[i] + kConfThreads - 1) / kConfThreads;
            p->rows_per_exp_blk[i] = g_fill_f4_per_block;                   // float4 per block
            const size_t totalq = (size_t)p->cells[i] * CH / 4;
            p->exp_nblk[i] = (int)((totalq + g_fill_f4_per_block - 1) / g_fill_f4_per_block);
            if (p->exp_nblk[i] < 1) p->exp_nblk[i] = 1;
        }
        p->rec_off[i] = off;      off += align_up(nT * 4 * sizeof(float), 256);
        p->tcell_ws_off[i] = off; off += align_up(nT * sizeof(int), 256);
        p->trank_off[i] = off;    off += align_up(nT * sizeof(int), 256);
        p->bits_off[i] = off;     off += align_up(nT * p->W * sizeof(unsigned), 256);
        p->counts_off[i] = off;   off += 256;
        p->confp_off[i] = off;    off += align_up((size_t)p->conf_nblk[i] * 2 * sizeof(float), 256);
        p->objp_off[i] = off;     off += align_up(nT * 8 * sizeof(float), 256);
        p->gconf_off[i] = coff;   coff += align_up((size_t)p->cells[i] * sizeof(float), 256);
        p->gobj_off[i] = coff;    coff += align_up(nT * CH * sizeof(float), 256);
        p->tcell_off[i] = coff;   coff += align_up(nT * sizeof(int), 256);
    }
    p->ws_total = off;
    p->compact_total = coff;
    return true;
}

bool any_grad(const odpl_yolo_loss_desc_t* d) {
    for (int i = 0; i < d->nscales; ++i) if (d->scale[i].grad) return true;
    return false;
}

// fills the kernel parameter block; `ws` is NULL in backward, `compact` is NULL in a forward-only call
int make_params(const odpl_yolo_loss_desc_t* d, const Plan& pl, char* ws, char* compact, Params* out) {
    Params& P = *out;
    P.target = d->target; P.upstream = nullptr;
    P.nscales = d->nscales; P.B = d->B; P.C = d->C; P.CH = 5 + d->C; P.nT = d->nT; P.W = pl.W;
    P.coord_sl1 = d->coord_mode == ODPL_COORD_SMOOTH_L1; P.cls_mode = d->cls_mode; P.average = d->average;
    P.ignore_thres = d->ignore_thres; P.obj_scale = d->obj_scale; P.noobj_scale = d->noobj_scale;
    P.inv_S = d->average ? 1.0f / (float)d->nscales : 1.0f;
    P.fill_nt = g_fill_nt;
    P.stamp = g_stamp;
    P.out7 = nullptr; P.out_scales = nullptr; P.out_counts = nullptr;
    for (int i = 0; i < d->nscales; ++i) {
        const odpl_yolo_scale_t& s = d->scale[i];
        Scale& S = P.s[i];
        if (((uintptr_t)s.head & 15) != 0 || ((uintptr_t)s.grad & 15) != 0) return ODPL_EINVAL;
        S.head = s.head; S.grad = s.grad; S.G = s.G; S.G2 = s.G * s.G; S.A = s.A; S.layout = pl.layout[i]; S.cells = pl.cells[i];
        S.inv_ch = ((1ull << 40) + (unsigned long long)P.CH - 1ull) / (unsigned long long)P.CH;
        if (ws) {
            S.state = (int*)(ws + pl.state_off[i]); S.rec = (float*)(ws + pl.rec_off[i]);
            S.trank = (int*)(ws + pl.trank_off[i]); S.bits = (unsigned*)(ws + pl.bits_off[i]);
            S.counts = (int*)(ws + pl.counts_off[i]);
            S.conf_part = (float*)(ws + pl.confp_off[i]); S.obj_part = (float*)(ws + pl.objp_off[i]);
        } else {
            S.state = nullptr; S.rec = nullptr; S.trank = nullptr; S.bits = nullptr; S.counts = nullptr; S.conf_part = nullptr; S.obj_part = nullptr;
        }
        if (compact) {
            S.gconf = (float*)(compact + pl.gconf_off[i]); S.gobj = (float*)(compact + pl.gobj_off[i]);
            S.tcell = (int*)(compact + pl.tcell_off[i]);
        } else {
            S.gconf = nullptr; S.gobj = nullptr; S.tcell = ws ? (int*)(ws + pl.tcell_ws_off[i]) : nullptr;
        }
        for (int a = 0; a < ODPL_MAX_ANCHORS; ++a) { S.aw[a] = s.anchor_w[a]; S.ah[a] = s.anchor_h[a]; }
        S.conf_blk_per_row = pl.conf_blk_per_row[i]; S.conf_nblk = pl.conf_nblk[i]; S.conf_vec = pl.conf_vec[i];
        S.exp_nblk = pl.exp_nblk[i]; S.rows_per_exp_blk = pl.rows_per_exp_blk[i];
        S.conf_blk0 = S.obj_blk0 = S.exp_blk0 = 0;
    }
    return 0;
}

int run_forward(const odpl_yolo_loss_desc_t* d, const Plan& pl, void* workspace, void* compact, float* out7, float* out_scales,
                int32_t* out_counts, hipStream_t stream) {
    Params P;
    const int rc = make_params(d, pl, (char*)workspace, (char*)compact, &P);
    if (rc) return rc;
    P.out7 = out7; P.out_scales = out_scales; P.out_counts = out_counts;
    int nblk = 0;
    for (int i = 0; i < d->nscales; ++i) { P.s[i].conf_blk0 = nblk; nblk += P.s[i].conf_nblk; }     // confidence blocks first
    for (int i = 0; i < d->nscales; ++i) { P.s[i].obj_blk0 = nblk; nblk += d->nT; }
    if (d->nT <= g_fused_max_t && d->nT <= kFusedMaxT && d->C <= 256) {      // one launch, no state map (left as it is: still clean if it was)
        prof_begin(ODPL_PROF_YOLO_LOSS, stream);
        hipLaunchKernelGGL(yolo_fused_kernel, dim3(nblk), dim3(kConfThreads), 0, stream, P);
        prof_end(ODPL_PROF_YOLO_LOSS, stream);
        ODPL_HIP_TRY(hipGetLastError());
        prof_begin(ODPL_PROF_YOLO_FINALIZE, stream);
        hipLaunchKernelGGL(yolo_finalize_kernel, dim3(1), dim3(64 * ODPL_MAX_SCALES), 0, stream, P);
        prof_end(ODPL_PROF_YOLO_FINALIZE, stream);
        ODPL_HIP_TRY(hipGetLastError());
        return 0;
    }
    if (!(d->flags & ODPL_YOLO_STATE_CLEAN)) ODPL_HIP_TRY(odpl::fill_bytes(workspace, 0xFF, pl.state_bytes_total, stream));
    prof_begin(ODPL_PROF_YOLO_ASSIGN, stream);
    hipLaunchKernelGGL(yolo_assign_kernel, dim3(d->nscales), dim3(kAssignThreads), 0, stream, P);
    prof_end(ODPL_PROF_YOLO_ASSIGN, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_YOLO_LOSS, stream);
    hipLaunchKernelGGL(yolo_loss_kernel, dim3(nblk), dim3(kConfThreads), 0, stream, P);
    prof_end(ODPL_PROF_YOLO_LOSS, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_YOLO_FINALIZE, stream);
    hipLaunchKernelGGL(yolo_finalize_kernel, dim3(1), dim3(64 * ODPL_MAX_SCALES), 0, stream, P);
    prof_end(ODPL_PROF_YOLO_FINALIZE, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

int run_backward(const odpl_yolo_loss_desc_t* d, const Plan& pl, const void* compact, const float* upstream, hipStream_t stream) {
    Params P;
    const int rc = make_params(d, pl, nullptr, (char*)compact, &P);
    if (rc) return rc;
    P.upstream = upstream;
    int nblk = 0;
    for (int i = 0; i < d->nscales; ++i) {
        P.s[i].exp_blk0 = nblk;
        if (!d->scale[i].grad) P.s[i].exp_nblk = 0;
        nblk += P.s[i].exp_nblk;
    }
    if (nblk == 0) return 0;
    prof_begin(ODPL_PROF_YOLO_EXPAND, stream);
    hipLaunchKernelGGL(yolo_expand_kernel, dim3(nblk), dim3(kExpandThreads), 0, stream, P);
    prof_end(ODPL_PROF_YOLO_EXPAND, stream);
    ODPL_HIP_TRY(hipGetLastError());
    if (d->nT > 0) {
        prof_begin(ODPL_PROF_YOLO_PATCH, stream);
        hipLaunchKernelGGL(yolo_patch_kernel, dim3(d->nT, d->nscales), dim3(kObjThreads), 0, stream, P);
        prof_end(ODPL_PROF_YOLO_PATCH, stream);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

extern "C" int odpl_abi_version(void) { return ODPL_ABI_VERSION; }

extern "C" size_t odpl_yolo_region_loss_workspace_bytes(const odpl_yolo_loss_desc_t* desc) {
    Plan p;
    if (!make_plan(desc, &p)) return 0;
    return p.ws_total + (any_grad(desc) ? p.compact_total : 0);      // the one-call entry keeps the compact form in the workspace
}

extern "C" size_t odpl_yolo_region_loss_compact_bytes(const odpl_yolo_loss_desc_t* desc) {
    Plan p;
    return make_plan(desc, &p) ? p.compact_total : 0;
}

extern "C" int odpl_yolo_region_loss_forward(const odpl_yolo_loss_desc_t* d, void* workspace, size_t workspace_bytes,
                                             void* compact, size_t compact_bytes, float* out7, float* out_scales,
                                             int32_t* out_counts, void* stream) {
    Plan pl;
    if (!make_plan(d, &pl) || !workspace || workspace_bytes < pl.ws_total || !out7) return ODPL_EINVAL;
    if (compact && compact_bytes < pl.compact_total) return ODPL_EINVAL;
    if (d->nT > 0 && !d->target) return ODPL_EINVAL;
    if (((uintptr_t)workspace & 15) != 0 || ((uintptr_t)compact & 15) != 0) return ODPL_EINVAL;
    return run_forward(d, pl, workspace, compact, out7, out_scales, out_counts, (hipStream_t)stream);
}

extern "C" int odpl_yolo_region_loss_backward(const odpl_yolo_loss_desc_t* d, const void* compact, size_t compact_bytes,
                                              const float* upstream, void* stream) {
    Plan pl;
    if (!make_plan(d, &pl) || !compact || compact_bytes < pl.compact_total || ((uintptr_t)compact & 15) != 0) return ODPL_EINVAL;
    return run_backward(d, pl, compact, upstream, (hipStream_t)stream);
}

extern "C" int odpl_yolo_region_loss(const odpl_yolo_loss_desc_t* d, void* workspace, size_t workspace_bytes,
                                     float* out7, float* out_scales, int32_t* out_counts, void* stream_) {
    Plan pl;
    if (!make_plan(d, &pl) || !workspace || !out7) return ODPL_EINVAL;
    const bool grads = any_grad(d);
    if (workspace_bytes < pl.ws_total + (grads ? pl.compact_total : 0)) return ODPL_EINVAL;
    if (d->nT > 0 && !d->target) return ODPL_EINVAL;
    if (((uintptr_t)workspace & 15) != 0) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* compact = grads ? (char*)workspace + pl.ws_total : nullptr;            // every offset is a multiple of 256
    const int rc = run_forward(d, pl, workspace, compact, out7, out_scales, out_counts, stream);
    if (rc || !grads) return rc;
    return run_backward(d, pl, compact, nullptr, stream);
}

extern "C" int odpl_yolo_compact_view(const odpl_yolo_loss_desc_t* d, const void* compact, int32_t scale, odpl_sparse_grad_t* out) {
    Plan pl;
    if (!make_plan(d, &pl) || !compact || !out || scale < 0 || scale >= d->nscales) return ODPL_EINVAL;
    const char* c = (const char*)compact;
    out->dense = (const float*)(c + pl.gconf_off[scale]);
    out->rows = (const float*)(c + pl.gobj_off[scale]);
    out->row_cell = (const int32_t*)(c + pl.tcell_off[scale]);
    out->n_rows = d->nT; out->n_rows_dev = nullptr;
    out->B = d->B; out->G2 = d->scale[scale].G * d->scale[scale].G; out->A = d->scale[scale].A; out->no = 5 + d->C;
    out->dense_ch = 4;
    out->layout = pl.layout[scale];
    return 0;
}

namespace odpl { extern int g_retina_fill_nt; extern int g_bn_target_blocks; extern int g_bn_stream_max_blocks; extern int g_bn_max_rows_per_tile; extern int g_bn_slice_from_c; }

extern "C" int odpl_debug_set_stamp_buffer(void* dev_u64x2) { g_stamp = (unsigned long long*)dev_u64x2; return 0; }

extern "C" int odpl_tune_set(int32_t key, int32_t value) {
    switch (key) {
        case ODPL_TUNE_RETINA_FILL_NT: odpl::g_retina_fill_nt = value ? 1 : 0; return 0;
        case ODPL_TUNE_YOLO_FILL_F4_PER_BLOCK: if (value < 64 || value > (1 << 20)) return ODPL_EINVAL; g_fill_f4_per_block = value; return 0;
        case ODPL_TUNE_YOLO_FILL_NT: g_fill_nt = value ? 1 : 0; return 0;
        case ODPL_TUNE_BN_TARGET_BLOCKS: if (value < 64 || value > (1 << 20)) return ODPL_EINVAL; odpl::g_bn_target_blocks = value; return 0;
        case ODPL_TUNE_BN_STREAM_MAX_BLOCKS: if (value != 0 && (value < 64 || value > (1 << 20))) return ODPL_EINVAL; odpl::g_bn_stream_max_blocks = value; return 0;
        case ODPL_TUNE_BN_MAX_ROWS_PER_TILE: if (value < 16 || value > (1 << 16)) return ODPL_EINVAL; odpl::g_bn_max_rows_per_tile = value; return 0;
        case ODPL_TUNE_BN_SLICE_FROM_C: if (value < 4) return ODPL_EINVAL; odpl::g_bn_slice_from_c = value; return 0;
        case ODPL_TUNE_YOLO_FUSED_MAX_TARGETS: if (value < 0) return ODPL_EINVAL; g_fused_max_t = value > kFusedMaxT ? kFusedMaxT : value; return 0;
        default: return ODPL_EINVAL;
    }
}

extern "C" int odpl_scale_inplace_if_not_one(float* grad, size_t n, const float* scale, void* stream_) {
    if (!grad || !scale || ((uintptr_t)grad & 15) != 0) return ODPL_EINVAL;
    if (n == 0) return 0;
    size_t blocks = (n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(scale_if_not_one_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, grad, n, scale);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_scale_inplace_if_not_one_bf16(void* grad, size_t n, const float* scale, void* stream_) {
    if (!grad || !scale) return ODPL_EINVAL;
    if (n == 0) return 0;
    size_t blocks = (n + 255) / 256; if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(scale_if_not_one_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream_, (__hip_bfloat16*)grad, n, scale);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
