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
constexpr int kT = 256;           // anchors per tile / workgroup
constexpr int kRowWGs = 512;      // workgroups of the main launch that walk the positive rows (4 waves each)
}
namespace odpl { int g_retina_fill_nt = 1; }
namespace {

struct RParams {
    const float* loc; const float* cls; float* grad_loc; float* grad_cls;
    const float* anchors; const float* target;
    int B, An, C, nT, coord_sl1, nblk_x, ntiles;
    float img_size, img_size_y;       // target scale of the x / w columns and of the y / h columns (equal in the reference)
    int* code;                     // [B*An]  -1 ignore, 0 background, (target_row << 12 | label+1) positive
    int* tile_cnt;                 // [ntiles] positives per tile
    int* tile_off;                 // [ntiles+1] exclusive scan; tile_off[ntiles] = num_pos
    int* row_tile;                 // [B*An] (first num_pos used): tile of positive row r
    unsigned char* tile_pos;       // [ntiles*256] (first tile_cnt used): local index of the k-th positive of a tile
    int* status;                   // [0] status bits (kept 0 between calls)
    float* part;                   // [B*An*2] (first num_pos*2 used): loc term, cls term per positive row
    float* out3; int* out_counts;
    int* pos_anchor; int* num_pos_out;   // ODPL_RETINA_ROWS_ONLY: the positive rows (b * An + a, ascending) and their number
};

// ---- assign: thread = anchor; the image's boxes come straight from the target rows, compacted in row
// order through LDS 256 rows at a time (losses.py:425-441)
struct GtLds { float x1, y1, x2, y2, area; int row; int label; };

__global__ __launch_bounds__(kT) void retina_assign_kernel(const RParams P) {
    __shared__ GtLds s_gt[kT];
    __shared__ int s_wcnt[kT / 64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, a = blockIdx.x * kT + tid;
    const bool live = a < P.An;
    Box4 ab; float a_area = 0.f;
    if (live) {
        const float4 an = reinterpret_cast<const float4*>(P.anchors)[a];
        ab.x1 = an.x - an.z / 2.0f; ab.y1 = an.y - an.w / 2.0f;          // change_box_order, losses.py:373
        ab.x2 = an.x + an.z / 2.0f; ab.y2 = an.y + an.w / 2.0f;
        a_area = (ab.x2 - ab.x1 + 1.0f) * (ab.y2 - ab.y1 + 1.0f);        // losses.py:400
    }
    float best = 0.f; int best_row = -1, best_label = 0, M = 0;
    for (int base = 0; base < P.nT; base += kT) {
        const int t = base + tid;
        bool match = false;
        float y1 = 0, y2 = 0, y3 = 0, y4 = 0, y5 = 0;
        if (t < P.nT) {
            const float* y = P.target + (size_t)t * 6;
            match = (y[0] == (float)b);
            y1 = y[1]; y2 = y[2]; y3 = y[3]; y4 = y[4]; y5 = y[5];
        }
        const unsigned long long mk = __ballot(match);
        __syncthreads();                                   // previous chunk fully consumed
        if (lane == 0) s_wcnt[wv] = __popcll(mk);
        __syncthreads();
        int off = 0, n = 0;
        for (int w = 0; w < kT / 64; ++w) { if (w < wv) off += s_wcnt[w]; n += s_wcnt[w]; }
        if (match) {
            GtLds r;
            const float cx = y2 * P.img_size, cy = y3 * P.img_size_y, w = y4 * P.img_size, h = y5 * P.img_size_y;
            r.x1 = cx - w / 2.0f; r.y1 = cy - h / 2.0f; r.x2 = cx + w / 2.0f; r.y2 = cy + h / 2.0f;
            r.area = (r.x2 - r.x1 + 1.0f) * (r.y2 - r.y1 + 1.0f);         // losses.py:401
            r.row = t; r.label = (int)trunc_ll(y1);
            s_gt[off + __popcll(mk & ((1ull << lane) - 1ull))] = r;
        }
        __syncthreads();
        if (live) {
            for (int j = 0; j < n; ++j) {
                const GtLds& r = s_gt[j];
                // losses.py:394-402: lt = max, rb = min, wh = (rb - lt + 1).clamp(0)
                const float w = fmaxf(fminf(ab.x2, r.x2) - fmaxf(ab.x1, r.x1) + 1.0f, 0.0f);
                const float h = fmaxf(fminf(ab.y2, r.y2) - fmaxf(ab.y1, r.y1) + 1.0f, 0.0f);
                const float inter = w * h;
                const float v = inter / (a_area + r.area - inter);
                if (M + j == 0 || (best == best && (v > best || v != v))) {   // first index on ties
                    best = v; best_row = r.row; best_label = r.label;
                }
            }
        }
        M += n;
    }
    int code = 0;
    if (live && M > 0) {
        const int cls_t = 1 + best_label;                                  // losses.py:437
        code = best < 0.5f ? 0 : cls_t;                                    // :439
        if (best > 0.4f && best < 0.5f) code = -1;                         // :440-441
        if (code > 0) {
            if (code > P.C) { atomicOr(&P.status[0], 2); code = 0; }       // eye(C+1)[y] IndexError upstream
            else code = (best_row << 12) | code;
        }
    }
    if (M == 0 && tid == 0) atomicOr(&P.status[0], 1);                     // image without boxes: reference raises
    if (live) P.code[(size_t)b * P.An + a] = code;
    // positives of this tile, ascending local index
    const unsigned long long pk = __ballot(code > 0);
    __syncthreads();
    if (lane == 0) s_wcnt[wv] = __popcll(pk);
    __syncthreads();
    int off = 0, n = 0;
    for (int w = 0; w < kT / 64; ++w) { if (w < wv) off += s_wcnt[w]; n += s_wcnt[w]; }
    const int tile = b * P.nblk_x + blockIdx.x;
    if (code > 0) P.tile_pos[(size_t)tile * kT + off + __popcll(pk & ((1ull << lane) - 1ull))] = (unsigned char)tid;
    if (tid == 0) P.tile_cnt[tile] = n;
}

// ---- scan: exclusive scan of the per-tile counts (1 WG), num_pos, and the row -> tile map
__global__ __launch_bounds__(1024) void retina_scan_kernel(const RParams P) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < P.ntiles; base += 1024) {
        const int i = base + tid;
        const int c = i < P.ntiles ? P.tile_cnt[i] : 0;
        int v = c;                                          // inclusive wave scan
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
        if (lane == 63) s_w[wv] = v;
        __syncthreads();
        int woff = 0;
        for (int w = 0; w < wv; ++w) woff += s_w[w];
        const int excl = s_carry + woff + v - c;
        if (i < P.ntiles) {
            P.tile_off[i] = excl;
            for (int k = 0; k < c; ++k) P.row_tile[excl + k] = i;
        }
        __syncthreads();
        if (tid == 1023) s_carry = excl + c;
        __syncthreads();
    }
    if (tid == 0) P.tile_off[P.ntiles] = s_carry;
}

// ---- main: WGs [0, kRowWGs) walk the positive rows (one wave per row); the rest zero-fill the tiles
template <bool NT>
__global__ __launch_bounds__(kT) void retina_main_kernel(const RParams P) {
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, C = P.C;
    const bool want_grad = P.grad_cls != nullptr;
    if ((int)blockIdx.x < kRowWGs) {
        const int num_pos = P.tile_off[P.ntiles];
        const float npos = fmaxf(1.0f, (float)num_pos);                    // losses.py:469
        const float inv = 1.0f / npos;
        for (int r = blockIdx.x * (kT / 64) + wv; r < num_pos; r += kRowWGs * (kT / 64)) {
            const int tile = P.row_tile[r];
            const int local = P.tile_pos[(size_t)tile * kT + (r - P.tile_off[tile])];
            const int b = tile / P.nblk_x, a = (tile - b * P.nblk_x) * kT + local;
            const size_t ga = (size_t)b * P.An + a;
            if (P.pos_anchor && lane == 0) P.pos_anchor[r] = (int)ga;          // rows are walked in ascending (b, a) order of r
            const int cd = P.code[ga];
            const int trow = cd >> 12, label = (cd & 0xfff) - 1;
            const float* y = P.target + (size_t)trow * 6;
            const float4 an = reinterpret_cast<const float4*>(P.anchors)[a];
            // ---- localisation (losses.py:434-436, 458-461), lanes 0..3
            float l_loc = 0.f;
            if (lane < 4) {
                float tgt;
                if (lane == 0) tgt = (y[2] * P.img_size - an.x) / an.z;
                else if (lane == 1) tgt = (y[3] * P.img_size_y - an.y) / an.w;
                else if (lane == 2) tgt = logf(y[4] * P.img_size / an.z);
                else tgt = logf(y[5] * P.img_size_y / an.w);
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
            if (lane == 0) { P.part[(size_t)r * 2] = l_loc; P.part[(size_t)r * 2 + 1] = l_cls; }
        }
        return;
    }
    if (!want_grad) return;
    // ---- zero fill of one tile; tiles that own positive rows leave those rows to the row waves
    const int tile = blockIdx.x - kRowWGs;
    const int b = tile / P.nblk_x, a0 = (tile - b * P.nblk_x) * kT;
    const int na = (P.An - a0) < kT ? (P.An - a0) : kT;
    const size_t r0 = (size_t)b * P.An + a0;
    if (P.tile_cnt[tile] == 0) {
        fill_zero<NT, kT>(P.grad_cls, r0 * C, (r0 + na) * C);
        fill_zero<NT, kT>(P.grad_loc, r0 * 4, (r0 + na) * 4);
    } else {
        // the runs of rows BETWEEN the tile's positive rows (ascending local indices in tile_pos) are 16-byte store streams
        // like the clean tiles; only the positive rows themselves are left to the row waves
        if (tid < na && P.code[r0 + tid] <= 0) reinterpret_cast<float4*>(P.grad_loc)[r0 + tid] = make_float4(0.f, 0.f, 0.f, 0.f);
        const int cnt = P.tile_cnt[tile];
        const unsigned char* pos = P.tile_pos + (size_t)tile * kT;
        int start = 0;
        for (int i = 0; i <= cnt; ++i) {
            const int end = i < cnt ? (int)pos[i] : na;
            if (end > start) fill_zero<NT, kT>(P.grad_cls, (r0 + start) * C, (r0 + end) * C);
            start = end + 1;
        }
    }
}

// ODPL_RETINA_ASSIGN_ONLY: the positive rows in ascending (b, a) order and their number, before any prediction exists
__global__ __launch_bounds__(kT) void retina_poslist_kernel(const RParams P) {
    const int num_pos = P.tile_off[P.ntiles];
    for (int r = blockIdx.x * kT + threadIdx.x; r < num_pos; r += gridDim.x * kT) {
        const int tile = P.row_tile[r];
        const int local = P.tile_pos[(size_t)tile * kT + (r - P.tile_off[tile])];
        const int b = tile / P.nblk_x, a = (tile - b * P.nblk_x) * kT + local;
        P.pos_anchor[r] = b * P.An + a;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) P.num_pos_out[0] = num_pos;
}

__global__ __launch_bounds__(256) void retina_finalize_kernel(const RParams P) {
    __shared__ double sh[2 * 4];
    const int tid = threadIdx.x, n = P.tile_off[P.ntiles];
    double a = 0, c = 0;
    for (int i = tid; i < n; i += 256) { a += (double)P.part[(size_t)i * 2]; c += (double)P.part[(size_t)i * 2 + 1]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); c += __shfl_down(c, o, 64); }
    if ((tid & 63) == 0) { sh[(tid >> 6) * 2] = a; sh[(tid >> 6) * 2 + 1] = c; }
    __syncthreads();
    if (tid == 0) {
        const float loc = (float)(sh[0] + sh[2] + sh[4] + sh[6]), cls = (float)(sh[1] + sh[3] + sh[5] + sh[7]);
        const int st = P.status[0];
        const float npos = fmaxf(1.0f, (float)n);
        const bool bad = st & 1;                              // an image without boxes: the reference raises
        P.out3[0] = bad ? NAN : (loc + cls) / npos;           // losses.py:471
        P.out3[1] = bad ? NAN : loc / npos;
        P.out3[2] = bad ? NAN : cls / npos;
        if (P.out_counts) { P.out_counts[0] = n; P.out_counts[1] = st; P.out_counts[2] = 0; P.out_counts[3] = 0; }
        if (P.num_pos_out) P.num_pos_out[0] = n;
        P.status[0] = 0;                                      // clean for the next call
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct RPlan { size_t status, code, tile_cnt, tile_off, row_tile, tile_pos, part, total; int nblk_x, ntiles; };

bool make_rplan(const odpl_retina_loss_desc_t* d, RPlan* p) {
    if (!d || d->B < 1 || d->An < 1 || d->C < 1 || d->C > 4094 || d->nT < 0 || d->nT > (1 << 19)) return false;
    const bool assign_only = (d->flags & ODPL_RETINA_ASSIGN_ONLY) != 0;
    if (assign_only && (d->flags & ODPL_RETINA_ASSIGNED)) return false;
    if (!d->anchors || (d->nT > 0 && !d->target)) return false;
    if (assign_only) {
        if (!d->pos_anchor || !d->num_pos) return false;
    } else {
        if (!d->loc || !d->cls) return false;
        if ((d->grad_loc == nullptr) != (d->grad_cls == nullptr)) return false;
        if ((d->flags & ODPL_RETINA_ROWS_ONLY) && (!d->grad_cls || !d->pos_anchor || !d->num_pos)) return false;
    }
    if ((long long)d->B * d->An * d->C > 0x7fffffffLL * 4) return false;
    p->nblk_x = (d->An + kT - 1) / kT;
    p->ntiles = p->nblk_x * d->B;
    const size_t cells = (size_t)d->B * d->An;
    size_t off = 0;
    p->status = off;   off += 256;              // first: kept zero between calls
    p->code = off;     off += align_up(cells * sizeof(int), 256);
    p->tile_cnt = off; off += align_up((size_t)p->ntiles * sizeof(int), 256);
    p->tile_off = off; off += align_up(((size_t)p->ntiles + 1) * sizeof(int), 256);
    p->row_tile = off; off += align_up(cells * sizeof(int), 256);
    p->tile_pos = off; off += align_up((size_t)p->ntiles * kT, 256);
    p->part = off;     off += align_up(cells * 2 * sizeof(float), 256);
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
    if (!make_rplan(d, &pl) || !workspace || workspace_bytes < pl.total) return ODPL_EINVAL;
    const bool assign_only = (d->flags & ODPL_RETINA_ASSIGN_ONLY) != 0, assigned = (d->flags & ODPL_RETINA_ASSIGNED) != 0;
    if (!assign_only && !out3) return ODPL_EINVAL;
    const uintptr_t al = (uintptr_t)workspace | (uintptr_t)d->anchors | (uintptr_t)d->grad_loc | (uintptr_t)d->grad_cls;
    if (al & 15) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    char* ws = (char*)workspace;
    RParams P;
    P.loc = d->loc; P.cls = d->cls; P.grad_loc = d->grad_loc; P.grad_cls = d->grad_cls;
    P.anchors = d->anchors; P.target = d->target;
    P.B = d->B; P.An = d->An; P.C = d->C; P.nT = d->nT; P.coord_sl1 = d->coord_mode == ODPL_COORD_SMOOTH_L1;
    P.nblk_x = pl.nblk_x; P.ntiles = pl.ntiles; P.img_size = d->img_size; P.img_size_y = d->img_size_y > 0.0f ? d->img_size_y : d->img_size;
    P.code = (int*)(ws + pl.code); P.tile_cnt = (int*)(ws + pl.tile_cnt); P.tile_off = (int*)(ws + pl.tile_off);
    P.row_tile = (int*)(ws + pl.row_tile); P.tile_pos = (unsigned char*)(ws + pl.tile_pos);
    P.status = (int*)(ws + pl.status); P.part = (float*)(ws + pl.part);
    P.out3 = out3; P.out_counts = out_counts;
    const bool rows_only = (d->flags & ODPL_RETINA_ROWS_ONLY) != 0;
    P.pos_anchor = (rows_only || assign_only) ? d->pos_anchor : nullptr; P.num_pos_out = (rows_only || assign_only) ? d->num_pos : nullptr;
    if (!assigned) {                                  // (an ASSIGNED call finds the status word as the ASSIGN_ONLY call left it)
        if (!(d->flags & ODPL_RETINA_STATE_CLEAN)) ODPL_HIP_TRY(odpl::fill_bytes(P.status, 0, 256, stream));
        const dim3 grid(pl.nblk_x, d->B);
        prof_begin(ODPL_PROF_RETINA_ASSIGN, stream);
        hipLaunchKernelGGL(retina_assign_kernel, grid, dim3(kT), 0, stream, P);
        prof_end(ODPL_PROF_RETINA_ASSIGN, stream);
        ODPL_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(retina_scan_kernel, dim3(1), dim3(1024), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    if (assign_only) {
        hipLaunchKernelGGL(retina_poslist_kernel, dim3(32), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
        return 0;
    }
    const int nmain = kRowWGs + ((d->grad_cls && !rows_only) ? pl.ntiles : 0);      // rows only: no zero stream
    prof_begin(ODPL_PROF_RETINA_MAIN, stream);
    if (g_retina_fill_nt) hipLaunchKernelGGL(retina_main_kernel<true>, dim3(nmain), dim3(kT), 0, stream, P);
    else hipLaunchKernelGGL(retina_main_kernel<false>, dim3(nmain), dim3(kT), 0, stream, P);
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
