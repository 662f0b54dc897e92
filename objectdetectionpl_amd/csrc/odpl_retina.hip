// RetinaNet loss for MI355X (gfx950): max-IoU anchor assignment + SmoothL1(sum) on positives +
// SOFTMAX focal(sum) + d loss / d (loc, cls), replacing RetinaNetLoss.forward (losses.py:405-472),
// box_iou / change_box_order (losses.py:360-403) and focal_loss (losses.py:317-349).
//
// What the loss touches.  focal_loss one-hots with the background column dropped (losses.py:336-339), so
// a background anchor's row contributes exactly 0: loss and gradient live on the num_pos positive anchors
// only.  Per call:
//   gt      (1 WG)                 per-image GT lists in row order: corners (+1 areas), centre form, label
//   assign  (An/256 x B WGs)       thread = anchor; the image's GT boxes are staged through LDS in tiles of
//                                  256; "+1" IoU in the reference's op order, first-argmax, label code
//                                  {-1 ignore, 0 background, (m<<16)|(label+1)} -> int32 [B,An]; num_pos by
//                                  integer atomicAdd (exact, order-free)
//   main    (An/256 x B WGs)       streams zeros into the workgroup's [256, C] / [256, 4] gradient tiles
//                                  (HBM-write bound), then one WAVE per positive row: wave-wide max / sum
//                                  for the softmax, focal + SmoothL1 terms and their gradients over the zeros
//   finalize(1 WG)                 fixed-order sum of the per-WG partials, / max(1, num_pos)
#include "odpl_common.hpp"
#include "odpl_boxes_dev.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;           // anchors per workgroup
constexpr int kGtTile = 256;      // GT boxes staged per LDS tile

struct GtRec { float x1, y1, x2, y2, area, cx, cy, w, h; int label; };   // 40 B

struct RParams {
    const float* loc; const float* cls; float* grad_loc; float* grad_cls;
    const float* anchors; const float* target;
    int B, An, C, nT, coord_sl1, nblk_x;
    float img_size;
    GtRec* gt; int* gt_off;        // [nT], [B+1]
    int* code;                     // [B*An]
    int* counters;                 // [0] num_pos (kept 0 between calls), [1] status
    float* part;                   // [B*nblk_x*2] loc_sum, cls_sum
    float* out3; int* out_counts;
};

__global__ __launch_bounds__(256) void retina_gt_kernel(const RParams P) {
    // one thread per image: count, then fill in row order (losses.py:425-426)
    __shared__ int s_cnt[1024];
    const int tid = threadIdx.x;
    int bad = 0;
    for (int b = tid; b < P.B; b += 256) {
        int c = 0;
        for (int t = 0; t < P.nT; ++t) c += (P.target[(size_t)t * 6] == (float)b);
        s_cnt[b] = c;
        bad |= (c == 0);           // the reference raises on an image without boxes (max over an empty dim)
    }
    bad = __syncthreads_or(bad);
    if (tid == 0) {
        int o = 0;
        for (int b = 0; b < P.B; ++b) { P.gt_off[b] = o; o += s_cnt[b]; }
        P.gt_off[P.B] = o;
        if (bad) atomicOr(&P.counters[1], 1);
    }
    __syncthreads();
    for (int b = tid; b < P.B; b += 256) {
        int o = P.gt_off[b];
        for (int t = 0; t < P.nT; ++t) {
            const float* y = P.target + (size_t)t * 6;
            if (y[0] != (float)b) continue;
            GtRec r;
            r.cx = y[2] * P.img_size; r.cy = y[3] * P.img_size; r.w = y[4] * P.img_size; r.h = y[5] * P.img_size;
            r.x1 = r.cx - r.w / 2.0f; r.y1 = r.cy - r.h / 2.0f;            // change_box_order, losses.py:373
            r.x2 = r.cx + r.w / 2.0f; r.y2 = r.cy + r.h / 2.0f;
            r.area = (r.x2 - r.x1 + 1.0f) * (r.y2 - r.y1 + 1.0f);          // losses.py:401
            r.label = (int)trunc_ll(y[1]);
            P.gt[o++] = r;
        }
    }
}

__global__ __launch_bounds__(kT) void retina_assign_kernel(const RParams P) {
    __shared__ GtRec s_gt[kGtTile];
    __shared__ int s_red[kT / 64];
    const int b = blockIdx.y, tid = threadIdx.x, a = blockIdx.x * kT + tid;
    const bool live = a < P.An;
    Box4 ab; float a_area = 0.f;
    if (live) {
        const float4 an = reinterpret_cast<const float4*>(P.anchors)[a];
        ab.x1 = an.x - an.z / 2.0f; ab.y1 = an.y - an.w / 2.0f;
        ab.x2 = an.x + an.z / 2.0f; ab.y2 = an.y + an.w / 2.0f;
        a_area = (ab.x2 - ab.x1 + 1.0f) * (ab.y2 - ab.y1 + 1.0f);
    }
    const int g0 = P.gt_off[b], M = P.gt_off[b + 1] - g0;
    float best = 0.f; int bi = 0;
    for (int base = 0; base < M; base += kGtTile) {
        const int n = (M - base) < kGtTile ? (M - base) : kGtTile;
        __syncthreads();
        if (tid < n) s_gt[tid] = P.gt[g0 + base + tid];
        __syncthreads();
        if (live) {
            for (int j = 0; j < n; ++j) {
                const GtRec& r = s_gt[j];
                // losses.py:394-402: lt = max, rb = min, wh = (rb - lt + 1).clamp(0)
                const float w = fmaxf(fminf(ab.x2, r.x2) - fmaxf(ab.x1, r.x1) + 1.0f, 0.0f);
                const float h = fmaxf(fminf(ab.y2, r.y2) - fmaxf(ab.y1, r.y1) + 1.0f, 0.0f);
                const float inter = w * h;
                const float v = inter / (a_area + r.area - inter);
                const int m = base + j;
                if (m == 0) { best = v; bi = 0; }
                else if (best == best && (v > best || v != v)) { best = v; bi = m; }   // first index on ties
            }
        }
    }
    int code = 0, pos = 0;
    if (live && M > 0) {
        const int cls_t = 1 + P.gt[g0 + bi].label;                         // losses.py:437
        if (best < 0.5f) code = 0; else code = cls_t;                      // :439
        if (best > 0.4f && best < 0.5f) code = -1;                         // :440-441
        if (code > 0) {
            if (code > P.C || bi > 32767) { atomicOr(&P.counters[1], 2); code = 0; }   // eye(C+1)[y] IndexError upstream
            else { code = (bi << 16) | code; pos = 1; }
        }
    }
    if (live) P.code[(size_t)b * P.An + a] = code;
    pos = wave_sum_i(pos);
    if ((tid & 63) == 0) s_red[tid >> 6] = pos;
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < kT / 64; ++w) n += s_red[w];
        if (n) atomicAdd(&P.counters[0], n);
    }
}

template <bool NT>
__global__ __launch_bounds__(kT) void retina_main_kernel(const RParams P) {
    __shared__ int s_pos[kT];
    __shared__ int s_wcnt[kT / 64];
    __shared__ float s_part[2 * (kT / 64)];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int a0 = blockIdx.x * kT, na = (P.An - a0) < kT ? (P.An - a0) : kT;
    const int C = P.C;
    const bool want_grad = P.grad_cls != nullptr;
    if (want_grad) {
        const size_t r0 = (size_t)b * P.An + a0;
        fill_zero<NT, kT>(P.grad_cls, r0 * C, (r0 + na) * C);
        fill_zero<NT, kT>(P.grad_loc, r0 * 4, (r0 + na) * 4);
    }
    int code = 0;
    if (tid < na) code = P.code[(size_t)b * P.An + a0 + tid];
    // positive rows of the tile in ascending order (ballot + popcount prefix): fixed summation order
    const unsigned long long mk = __ballot(code > 0);
    if (lane == 0) s_wcnt[wv] = __popcll(mk);
    __syncthreads();                                          // also orders the zero stores before the row stores
    int off = 0, npos_blk = 0;
    for (int w = 0; w < kT / 64; ++w) { if (w < wv) off += s_wcnt[w]; npos_blk += s_wcnt[w]; }
    if (code > 0) s_pos[off + __popcll(mk & ((1ull << lane) - 1ull))] = tid;
    __syncthreads();
    float acc_loc = 0.f, acc_cls = 0.f;
    if (npos_blk > 0) {
        const float npos = fmaxf(1.0f, (float)P.counters[0]);  // losses.py:469
        const float inv = 1.0f / npos;
        const int g0 = P.gt_off[b];
        for (int i = wv; i < npos_blk; i += kT / 64) {
            const int row = s_pos[i];
            const size_t ga = (size_t)b * P.An + a0 + row;
            const int cd = P.code[ga];
            const int m = cd >> 16, label = (cd & 0xffff) - 1;
            const GtRec r = P.gt[g0 + m];
            const float4 an = reinterpret_cast<const float4*>(P.anchors)[a0 + row];
            // ---- localisation (losses.py:434-436, 458-461), lanes 0..3
            float l_loc = 0.f;
            if (lane < 4) {
                float tgt;
                if (lane == 0) tgt = (r.cx - an.x) / an.z;
                else if (lane == 1) tgt = (r.cy - an.y) / an.w;
                else if (lane == 2) tgt = logf(r.w / an.z);
                else tgt = logf(r.h / an.w);
                const float d = P.loc[ga * 4 + lane] - tgt;
                l_loc = coord_elem(d, P.coord_sl1);
                if (want_grad) P.grad_loc[ga * 4 + lane] = coord_deriv(d, P.coord_sl1) * inv;
            }
            // ---- softmax focal on the row (losses.py:342-345): wave-wide max and sum
            const float* x = P.cls + ga * C;
            float mx = -INFINITY;
            for (int k = lane; k < C; k += 64) mx = fmaxf(mx, x[k]);
            mx = wave_max(mx); mx = __shfl(mx, 0, 64);
            float se = 0.f;
            for (int k = lane; k < C; k += 64) se += expf(x[k] - mx);
            se = wave_sum(se); se = __shfl(se, 0, 64);
            const float pc_raw = expf(x[label] - mx) / se;
            const float pc = fminf(fmaxf(pc_raw, 1e-7f), 1.0f - 1e-7f);
            const float om = 1.0f - pc, lg = logf(pc);
            const float l_cls = 0.25f * (-1.0f * lg) * (om * om);
            // d/dpc of alpha * (-log pc) * (1-pc)^2; clamp passes the gradient inside [1e-7, 1-1e-7] only
            const bool pass = (pc_raw >= 1e-7f) && (pc_raw <= 1.0f - 1e-7f);
            const float g = pass ? 0.25f * (-(om * om) / pc + 2.0f * om * lg) : 0.0f;
            if (want_grad) {
                float* gx = P.grad_cls + ga * C;
                for (int k = lane; k < C; k += 64) {
                    const float pk = expf(x[k] - mx) / se;
                    gx[k] = g * pc_raw * ((k == label ? 1.0f : 0.0f) - pk) * inv;
                }
            }
            l_loc = wave_sum(l_loc);
            if (lane == 0) { acc_loc += l_loc; acc_cls += l_cls; }
        }
    }
    if (lane == 0) { s_part[wv * 2] = acc_loc; s_part[wv * 2 + 1] = acc_cls; }
    __syncthreads();
    if (tid == 0) {
        float a = 0.f, c = 0.f;
        for (int w = 0; w < kT / 64; ++w) { a += s_part[w * 2]; c += s_part[w * 2 + 1]; }
        const size_t pb = ((size_t)b * P.nblk_x + blockIdx.x) * 2;
        P.part[pb] = a; P.part[pb + 1] = c;
    }
}

__global__ __launch_bounds__(256) void retina_finalize_kernel(const RParams P) {
    __shared__ double sh[2 * 4];
    const int tid = threadIdx.x, n = P.B * P.nblk_x;
    double a = 0, c = 0;
    for (int i = tid; i < n; i += 256) { a += (double)P.part[(size_t)i * 2]; c += (double)P.part[(size_t)i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
    if ((tid & 63) == 0) { sh[(tid >> 6) * 2] = a; sh[(tid >> 6) * 2 + 1] = c; }
    __syncthreads();
    if (tid == 0) {
        const float loc = (float)(sh[0] + sh[2] + sh[4] + sh[6]), cls = (float)(sh[1] + sh[3] + sh[5] + sh[7]);
        const int np = P.counters[0], st = P.counters[1];
        const float npos = fmaxf(1.0f, (float)np);
        const bool bad = st & 1;                              // an image without boxes: the reference raises
        P.out3[0] = bad ? NAN : (loc + cls) / npos;           // losses.py:471
        P.out3[1] = bad ? NAN : loc / npos;
        P.out3[2] = bad ? NAN : cls / npos;
        if (P.out_counts) { P.out_counts[0] = np; P.out_counts[1] = st; P.out_counts[2] = P.gt_off[P.B]; P.out_counts[3] = 0; }
        P.counters[0] = 0; P.counters[1] = 0;                 // clean for the next call
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct RPlan { size_t counters, gt, gt_off, code, part, total; int nblk_x; };

bool make_rplan(const odpl_retina_loss_desc_t* d, RPlan* p) {
    if (!d || d->B < 1 || d->B > 1024 || d->An < 1 || d->C < 1 || d->C > 65535 || d->nT < 0) return false;
    if (!d->loc || !d->cls || !d->anchors || (d->nT > 0 && !d->target)) return false;
    if ((d->grad_loc == nullptr) != (d->grad_cls == nullptr)) return false;
    p->nblk_x = (d->An + kT - 1) / kT;
    const size_t nT = d->nT > 0 ? d->nT : 1;
    size_t off = 0;
    p->counters = off; off += 256;              // first: kept zero between calls
    p->gt = off;       off += align_up(nT * sizeof(GtRec), 256);
    p->gt_off = off;   off += align_up(((size_t)d->B + 1) * sizeof(int), 256);
    p->code = off;     off += align_up((size_t)d->B * d->An * sizeof(int), 256);
    p->part = off;     off += align_up((size_t)d->B * p->nblk_x * 2 * sizeof(float), 256);
    p->total = off;
    return true;
}
}  // namespace

extern "C" size_t odpl_retina_loss_workspace_bytes(const odpl_retina_loss_desc_t* d) {
    RPlan p;
    return make_rplan(d, &p) ? p.total : 0;
}

extern "C" int odpl_retina_loss(const odpl_retina_loss_desc_t* d, void* workspace, size_t workspace_bytes,
                                float* out3, int32_t* out_counts, void* stream_) {
    RPlan pl;
    if (!make_rplan(d, &pl) || !workspace || workspace_bytes < pl.total || !out3) return ODPL_EINVAL;
    const uintptr_t al = (uintptr_t)workspace | (uintptr_t)d->anchors | (uintptr_t)d->grad_loc | (uintptr_t)d->grad_cls;
    if (al & 15) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    RParams P;
    P.loc = d->loc; P.cls = d->cls; P.grad_loc = d->grad_loc; P.grad_cls = d->grad_cls;
    P.anchors = d->anchors; P.target = d->target;
    P.B = d->B; P.An = d->An; P.C = d->C; P.nT = d->nT; P.coord_sl1 = d->coord_mode == ODPL_COORD_SMOOTH_L1;
    P.nblk_x = pl.nblk_x; P.img_size = d->img_size;
    P.gt = (GtRec*)(ws + pl.gt); P.gt_off = (int*)(ws + pl.gt_off); P.code = (int*)(ws + pl.code);
    P.counters = (int*)(ws + pl.counters); P.part = (float*)(ws + pl.part);
    P.out3 = out3; P.out_counts = out_counts;
    if (!(d->flags & ODPL_RETINA_STATE_CLEAN)) ODPL_HIP_TRY(hipMemsetAsync(P.counters, 0, 256, stream));
    hipLaunchKernelGGL(retina_gt_kernel, dim3(1), dim3(256), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    const dim3 grid(pl.nblk_x, d->B);
    prof_begin(ODPL_PROF_RETINA_ASSIGN, stream);
    hipLaunchKernelGGL(retina_assign_kernel, grid, dim3(kT), 0, stream, P);
    prof_end(ODPL_PROF_RETINA_ASSIGN, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_RETINA_MAIN, stream);
    hipLaunchKernelGGL(retina_main_kernel<true>, grid, dim3(kT), 0, stream, P);
    prof_end(ODPL_PROF_RETINA_MAIN, stream);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(retina_finalize_kernel, dim3(1), dim3(256), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------
namespace {
struct AnchorParams { int fm_w[5], fm_h[5], first[6]; float grid_w[5], grid_h[5]; float wh[5][9][2]; };

__global__ __launch_bounds__(256) void retina_anchors_kernel(const AnchorParams P, float* __restrict__ out, long long n) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int lv = 0;
    while (lv < 4 && i >= P.first[lv + 1]) ++lv;
    const int r = (int)(i - P.first[lv]);
    const int k = r % 9, cell = r / 9, x = cell % P.fm_w[lv], y = cell / P.fm_w[lv];
    float4 o;
    o.x = ((float)x + 0.5f) * P.grid_w[lv];                 // RetinaUtils.py:25-26
    o.y = ((float)y + 0.5f) * P.grid_h[lv];
    o.z = P.wh[lv][k][0]; o.w = P.wh[lv][k][1];
    reinterpret_cast<float4*>(out)[i] = o;
}
}  // namespace

extern "C" int odpl_retina_anchors(const int32_t* fm, const float* grid, const float* wh, float* out, int64_t n_out, void* stream_) {
    if (!fm || !grid || !wh || !out || ((uintptr_t)out & 15)) return ODPL_EINVAL;
    AnchorParams P;
    long long tot = 0;
    for (int i = 0; i < 5; ++i) {
        P.fm_w[i] = fm[i * 2]; P.fm_h[i] = fm[i * 2 + 1]; P.grid_w[i] = grid[i * 2]; P.grid_h[i] = grid[i * 2 + 1];
        if (P.fm_w[i] < 1 || P.fm_h[i] < 1) return ODPL_EINVAL;
        P.first[i] = (int)tot; tot += (long long)P.fm_w[i] * P.fm_h[i] * 9;
        for (int k = 0; k < 9; ++k) { P.wh[i][k][0] = wh[(i * 9 + k) * 2]; P.wh[i][k][1] = wh[(i * 9 + k) * 2 + 1]; }
    }
    P.first[5] = (int)tot;
    if (tot != n_out || tot > 0x7fffffffLL) return ODPL_EINVAL;
    hipLaunchKernelGGL(retina_anchors_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, P, out, tot);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
