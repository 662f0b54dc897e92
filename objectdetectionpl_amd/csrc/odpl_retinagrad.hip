// Backward of RetinaNet's two final head convolutions from the STRUCTURED gradient of RetinaNetLoss.
//
// `cls_head[-1]` / `loc_head[-1]` (model/RetinaNet.py:93-99: Conv2d(256, 9 * C, 3, 1, 1) and Conv2d(256, 9 * 4, 3, 1, 1), applied to the five
// FPN levels) feed the criterion directly: their outputs, permuted to [B, H, W, 9, no] and concatenated over the levels, ARE
// cls_preds / loc_preds (RetinaNet.py:101-113).  The focal one-hot drops the background column and SmoothL1 runs on positives only
// (losses.py:336-339, 458-461), so d loss/d (cls_preds, loc_preds) is zero except on the num_pos positive anchors -- 863 of 802 800 rows at
// BASELINE config 4.  A generic convolution backward multiplies the whole [4, 720, 100, 167] gradient anyway: 0.59 TFLOP, 7.2 ms of
// MIOpen kernels per step (profiles/r02_retina_steady_state.csv: the two largest wrw / bwd launches).  Here, per positive anchor
// p = (b, level, y, x, k) with gradient row g[0..no):
//     dW[k*no + c, :, ky, kx] += g[c] * X[b, y+ky-1, x+kx-1, :]          db[k*no + c] += g[c]
//     dX[b, y+ky-1, x+kx-1, :] += sum_c g[c] * W[k*no + c, :, ky, kx]     (zero everywhere else)
// Three launches for both heads and all levels, no float atomics (bitwise reproducible):
//   map   the positive rows arrive sorted by (image, anchor index) = (image, level, pixel, k): the rows of one pixel are consecutive.
//         One thread per row marks the first row of its pixel and counts the pixel's rows in an int32 map over all level pixels
//   pair  one WAVE per (positive row, tap): the row's contribution to ONE input pixel, sum_c g[c] * W[k*no + c, :, tap] for both heads, into a
//         staging slot (eight weight rows in flight, lanes along Cin): 7.8 k independent waves at config 4 instead of a few dozen
//         serial chains on the pixels that many rows reach
//   dx    one WAVE per level pixel: it looks at the <= 9 map entries of its 3 x 3 neighbourhood; nothing there -> a zero row (the dense
//         dX the previous layer's backward expects is written exactly once: an HBM write stream, 2 x 91 MB at config 4); otherwise the
//         staged contributions are summed in (tap, row) order
//   dw    one workgroup per (head, anchor k, four output channels, tap): the rows of anchor k are compacted into LDS in row order; wave w
//         takes rows w, w + 4, ... eight at a time (eight X vectors in flight, lanes along Cin); the four waves are combined in wave order
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;
constexpr int kMaxHeads = 2;
constexpr int kRowChunk = 512;          // positive rows a dw workgroup stages in LDS at a time
int g_stage_rows = 4096;                // positive rows with staging slots (9 taps x 2 heads x Cin floats each: 75 MB at Cin = 256); the rows
                                        // beyond are summed in place by the dx launch (slower, same arithmetic): BASELINE config 4 has 863

struct Level { int H, W, HW, first, pix0; };      // first: anchor index of the level's first anchor; pix0: its first pixel in the map (per image: + b * HW)
struct HeadP {
    const float* x[ODPL_RETINA_LEVELS]; float* gx[ODPL_RETINA_LEVELS];
    const float* w; float* gw; float* gb; const float* rows;
    long long ws_o, ws_i, ws_ky, ws_kx;
    int no, dw_blk0, dw_nblk, chunks;
};
struct GP {
    HeadP h[kMaxHeads]; Level lv[ODPL_RETINA_LEVELS];
    const int* pos; const int* num_pos; const float* upstream;
    int* map_first; int* map_cnt;
    float* stage;                                                // [cap * 9][kMaxHeads][Cin]: one row's contribution through one tap
    int B, A, Cin, cv, nlevels, nheads, An, npix_img, npix, cap; // npix_img = sum HW; npix = B * npix_img; cap: rows that own staging slots
    int pair_head;                                               // the head a pair launch works on
    int rows_an;                                                 // anchors per image in the heads' gradient tensors (An, or more: a level of a wider tensor)
    int shared_x;                                                // both heads read ONE input: their dX is summed into head 0's tensor
};

// anchor index a (within an image) -> level, cell, k
__device__ __forceinline__ void decode_anchor(const GP& P, int a, int& l, int& cell, int& k) {
    l = 0;
    for (int i = 1; i < P.nlevels; ++i) if (a >= P.lv[i].first) l = i;
    const int r = a - P.lv[l].first;
    cell = r / P.A; k = r - cell * P.A;
}
// map slot of (level l, image b, cell): pixels are laid out level-major, image-major inside a level
__device__ __forceinline__ int map_slot(const GP& P, int l, int b, int cell) { return P.B * P.lv[l].pix0 + b * P.lv[l].HW + cell; }

// row of anchor ga = b * An + a in a head's gradient tensor [B, rows_an, no]
__device__ __forceinline__ size_t grad_row(const GP& P, int ga) { const int b = ga / P.An; return (size_t)b * P.rows_an + (ga - b * P.An); }

__global__ __launch_bounds__(kT) void rh_map_kernel(const GP P) {
    const int n = *P.num_pos;
    for (int r = blockIdx.x * kT + threadIdx.x; r < n; r += gridDim.x * kT) {
        const int ga = P.pos[r], b = ga / P.An, a = ga - b * P.An;
        int l, cell, k; decode_anchor(P, a, l, cell, k);
        const int slot = map_slot(P, l, b, cell);
        bool first = r == 0;
        if (!first) {
            const int gp = P.pos[r - 1], bp = gp / P.An, ap = gp - bp * P.An;
            int lp, cp, kp; decode_anchor(P, ap, lp, cp, kp);
            first = map_slot(P, lp, bp, cp) != slot;
        }
        if (first) {
            int cnt = 1;
            for (; cnt < P.A && r + cnt < n; ++cnt) {                 // the pixel's rows follow (ascending k)
                const int gn = P.pos[r + cnt], bn = gn / P.An, an = gn - bn * P.An;
                int ln, cn, kn; decode_anchor(P, an, ln, cn, kn);
                if (map_slot(P, ln, bn, cn) != slot) break;
            }
            P.map_first[slot] = r; P.map_cnt[slot] = cnt;
        }
    }
}

__device__ __forceinline__ float4 ld_w4(const float* w, long long off, long long ws_i) {
    if (ws_i == 1) return *reinterpret_cast<const float4*>(w + off);
    return make_float4(w[off], w[off + ws_i], w[off + 2 * ws_i], w[off + 3 * ws_i]);
}

// sum_c g[c] * W[(k * no + c), 4v .. 4v+3, ky, kx] for one head: eight weight rows in flight
__device__ __forceinline__ float4 pair_contrib(const HeadP& Hd, const float* g, int k, int ky, int kx, int v) {
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    const long long base = (long long)k * Hd.no * Hd.ws_o + ky * Hd.ws_ky + kx * Hd.ws_kx + (long long)(4 * v) * Hd.ws_i;
    int c = 0;
    for (; c + 7 < Hd.no; c += 8) {
        float4 w[8]; float gc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { w[u] = ld_w4(Hd.w, base + (long long)(c + u) * Hd.ws_o, Hd.ws_i); gc[u] = g[c + u]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x += gc[u] * w[u].x; acc.y += gc[u] * w[u].y; acc.z += gc[u] * w[u].z; acc.w += gc[u] * w[u].w; }
    }
    for (; c < Hd.no; ++c) {
        const float4 w0 = ld_w4(Hd.w, base + (long long)c * Hd.ws_o, Hd.ws_i);
        const float g0 = g[c];
        acc.x += g0 * w0.x; acc.y += g0 * w0.y; acc.z += g0 * w0.z; acc.w += g0 * w0.w;
    }
    return acc;
}

// The staged contributions as small GEMMs.  For one (head, anchor k, tap) the weight slab W[k*no .. k*no+no, :, tap] (no x Cin floats:
// 80 KB for the class head) is the B operand of EVERY row of anchor k; one wave per (row, tap) re-read it per row -- 652 MB through L2s
// that hold 4 MB each.  Here a workgroup owns (head, k, tap, slot): it compacts the rows of anchor k (list order), loads the slab
// into LDS ONCE (in blocks of kSlabRows output channels) and walks the batches slot, slot + kPairSlots, ... of 16 rows: wave w takes
// rows w, w + 4, w + 8, w + 12 of a batch, lanes along Cin, gradients broadcast from LDS.  52 MB of weight reads at config 4.
constexpr int kPairSlots = 8;
constexpr int kSlabRows = 80;           // output channels of a slab block held in LDS (x Cin floats)
constexpr int kPairChunk = 512;         // rows of anchor k listed in LDS at a time

__global__ __launch_bounds__(kT) void rh_pair_kernel(const GP P) {
    __shared__ int s_ga[kPairChunk], s_r[kPairChunk];
    __shared__ int s_wave[kT / 64];
    extern __shared__ __align__(16) float s_dyn[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    int bid = blockIdx.x;
    const int slot = bid % kPairSlots; bid /= kPairSlots;
    const int t = bid % 9; bid /= 9;
    const int k = bid % P.A, h = P.pair_head;                           // one launch per head: the LDS slab is sized by that head's rows
    const HeadP& Hd = P.h[h];
    const int ky = t / 3, kx = t - 3 * ky;
    const int n = *P.num_pos, nlim = n < P.cap ? n : P.cap;
    const int cv = P.cv;
    const int slab = Hd.no < kSlabRows ? Hd.no : kSlabRows;
    float4* s_w = reinterpret_cast<float4*>(s_dyn);                     // [slab][cv]
    float* s_gb = s_dyn + (size_t)slab * cv * 4;                        // [16][kSlabRows]  gradients of the batch, channel block
    int seen = 0;                                                       // rows of anchor k listed so far (over the chunks)
    for (int r0 = 0; r0 < nlim; r0 += kPairChunk) {
        // ---- rows of anchor k in [r0, r0 + kPairChunk), ascending (ballot compaction)
        int base = 0;
        for (int t0 = r0; t0 < r0 + kPairChunk && t0 < nlim; t0 += kT) {
            const int r = t0 + tid;
            int ga = 0; bool keep = false;
            if (r < nlim && r < r0 + kPairChunk) {
                ga = P.pos[r];
                const int a = ga - (ga / P.An) * P.An;
                int l, cell, kk; decode_anchor(P, a, l, cell, kk);
                keep = kk == k;
            }
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_wave[wv] = __popcll(m);
            __syncthreads();
            int off = base, tot = 0;
            for (int w = 0; w < kT / 64; ++w) { if (w < wv) off += s_wave[w]; tot += s_wave[w]; }
            if (keep) { const int p = off + __popcll(m & ((1ull << lane) - 1ull)); s_ga[p] = ga; s_r[p] = r; }
            base += tot;
            __syncthreads();
        }
        const int cnt = base;
        // batches of this chunk that belong to this slot: global batch index (seen + i0) / 16
        for (int i0 = 0; i0 < cnt; i0 += 16) {
            if (((seen + i0) / 16) % kPairSlots != slot) continue;      // uniform
            const int nb = cnt - i0 < 16 ? cnt - i0 : 16;
            for (int v0 = 0; v0 < cv; v0 += 64) {
                const int v = v0 + lane;
                const bool ok = v < cv;
                float4 acc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) acc[u] = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int c0 = 0; c0 < Hd.no; c0 += kSlabRows) {
                    const int nc = Hd.no - c0 < kSlabRows ? Hd.no - c0 : kSlabRows;
                    __syncthreads();                                    // the previous block / batch is consumed
                    for (int e = tid; e < nc * cv; e += kT) {           // slab block: rows c0 .. c0 + nc of (k, tap), coalesced along Cin
                        const int c = e / cv, vv = e - c * cv;
                        s_w[c * cv + vv] = ld_w4(Hd.w, (long long)(k * Hd.no + c0 + c) * Hd.ws_o + ky * Hd.ws_ky + kx * Hd.ws_kx + (long long)(4 * vv) * Hd.ws_i, Hd.ws_i);
                    }
                    for (int e = tid; e < nb * nc; e += kT) {
                        const int u = e / nc, c = e - u * nc;
                        s_gb[u * kSlabRows + c] = Hd.rows[grad_row(P, s_ga[i0 + u]) * Hd.no + c0 + c];
                    }
                    __syncthreads();
                    if (ok) {
                        for (int c = 0; c < nc; ++c) {
                            const float4 w4 = s_w[c * cv + v];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const float g = s_gb[(wv + 4 * u) * kSlabRows + c];     // rows beyond nb read stale values: never stored
                                acc[u].x += g * w4.x; acc[u].y += g * w4.y; acc[u].z += g * w4.z; acc[u].w += g * w4.w;
                            }
                        }
                    }
                }
                if (ok) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = wv + 4 * u;
                        if (i < nb) {
                            const int ga = s_ga[i0 + i], b = ga / P.An, a = ga - b * P.An;
                            int l, cell, kk; decode_anchor(P, a, l, cell, kk);
                            const Level L = P.lv[l];
                            const int y = cell / L.W, x = cell - y * L.W, yy = y + ky - 1, xx = x + kx - 1;
                            if (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W)         // else: reaches no pixel, the slot is never read
                                reinterpret_cast<float4*>(P.stage)[((size_t)(s_r[i0 + i] * 9 + t) * kMaxHeads + h) * cv + v] = acc[u];
                        }
                    }
                }
            }
        }
        seen += cnt;
        __syncthreads();
    }
}

// The dense dX writer.  A wave owns kPixPerWave consecutive level pixels; lane 9 p + t looks up the map entry that reaches pixel p through
// tap t, so ONE trip to memory tells the wave which of its pixels anything reaches (almost none: those rows are streamed as zeros, 2 heads x
// Cin floats each); the others list their staging slots in (tap, row) order and sum them eight in flight per head.
constexpr int kPixPerWave = 7;

__global__ __launch_bounds__(kT) void rh_dx_kernel(const GP P) {
    __shared__ int s_list[kT / 64][96];                                // per wave: the staging slots (r * 9 + tap) that reach the current pixel
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pix0 = (blockIdx.x * (kT / 64) + wv) * kPixPerWave;      // wave-uniform
    if (pix0 >= P.npix) return;
    const float f = P.upstream ? *P.upstream : 1.0f;
    // ---- lane 9 p + t: map entry of the cell whose positives reach pixel pix0 + p through tap t
    int my_cnt = 0, my_fst = 0;
    {
        const int p = lane / 9, t = lane - 9 * p, pix = pix0 + p;
        if (p < kPixPerWave && pix < P.npix) {
            int l = 0;
            for (int i = 1; i < P.nlevels; ++i) if (pix >= P.B * P.lv[i].pix0) l = i;
            const Level L = P.lv[l];
            const int rel = pix - P.B * L.pix0, b = rel / L.HW, cell = rel - b * L.HW, y = cell / L.W, x = cell - y * L.W;
            const int ky = t / 3, kx = t - 3 * ky, yp = y - ky + 1, xp = x - kx + 1;
            if (yp >= 0 && yp < L.H && xp >= 0 && xp < L.W) {
                const int s = map_slot(P, l, b, yp * L.W + xp);
                my_cnt = P.map_cnt[s];
                if (my_cnt) my_fst = P.map_first[s];
            }
        }
    }
    const unsigned long long hit = __ballot(my_cnt != 0);
    int* list = s_list[wv];
    for (int p = 0; p < kPixPerWave; ++p) {
        const int pix = pix0 + p;
        if (pix >= P.npix) break;                                      // uniform
        int l = 0;
        for (int i = 1; i < P.nlevels; ++i) if (pix >= P.B * P.lv[i].pix0) l = i;
        const Level L = P.lv[l];
        const size_t row = (size_t)(pix - P.B * L.pix0) * P.cv;        // (b * HW + cell) * cv inside the level's tensor
        int M = 0; bool unstaged = false;
        if ((hit >> (9 * p)) & 0x1ffull) {                             // something reaches this pixel: list its slots in (tap, row) order
            for (int t = 0; t < 9; ++t) {
                const int m = __shfl(my_cnt, 9 * p + t, 64), f0 = __shfl(my_fst, 9 * p + t, 64);
                for (int j = 0; j < m && M < 96; ++j) { list[M++] = (f0 + j) * 9 + t; unstaged |= f0 + j >= P.cap; }
            }
        }
        for (int v0 = 0; v0 < P.cv; v0 += 64) {
            const int v = v0 + lane;
            const bool ok = v < P.cv;
            const int vc = ok ? v : P.cv - 1;
            float4 acc[kMaxHeads];
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (M && !unstaged) {
                // eight staged contributions per head in flight (unconditional loads clamped to the list's last entry), added in list order
                for (int i0 = 0; i0 < M; i0 += 8) {
                    float4 c4[kMaxHeads][8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int e = list[i0 + u < M ? i0 + u : M - 1];
#pragma unroll
                        for (int h = 0; h < kMaxHeads; ++h)
                            c4[h][u] = (h < P.nheads && P.h[h].gx[0]) ? reinterpret_cast<const float4*>(P.stage)[((size_t)e * kMaxHeads + h) * P.cv + vc] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u)
                        if (i0 + u < M) {
#pragma unroll
                            for (int h = 0; h < kMaxHeads; ++h) { acc[h].x += c4[h][u].x; acc[h].y += c4[h][u].y; acc[h].z += c4[h][u].z; acc[h].w += c4[h][u].w; }
                        }
                }
            } else if (M && ok) {
                for (int i = 0; i < M; ++i) {                          // more rows than staging slots: in place (same arithmetic per contribution)
                    const int e = list[i], r = e / 9, t = e - 9 * r, ky = t / 3, kx = t - 3 * ky;
#pragma unroll
                    for (int h = 0; h < kMaxHeads; ++h) {
                        if (h >= P.nheads || !P.h[h].gx[0]) continue;
                        float4 c4;
                        if (r < P.cap) c4 = reinterpret_cast<const float4*>(P.stage)[((size_t)e * kMaxHeads + h) * P.cv + v];
                        else {
                            const int ga = __builtin_amdgcn_readfirstlane(P.pos[r]);
                            const int a = ga - (ga / P.An) * P.An;
                            c4 = pair_contrib(P.h[h], P.h[h].rows + grad_row(P, ga) * P.h[h].no, (a - L.first) % P.A, ky, kx, v);
                        }
                        acc[h].x += c4.x; acc[h].y += c4.y; acc[h].z += c4.z; acc[h].w += c4.w;
                    }
                }
            }
            if (ok && P.shared_x) {                                     // both heads hang on one input (SSD's cl / bbx pairs): one dX, head 0 + head 1
                const v4f o = {(acc[0].x + acc[1].x) * f, (acc[0].y + acc[1].y) * f, (acc[0].z + acc[1].z) * f, (acc[0].w + acc[1].w) * f};
                __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(P.h[0].gx[l]) + row + v);
            } else if (ok) {
#pragma unroll
                for (int h = 0; h < kMaxHeads; ++h) {
                    if (h >= P.nheads || !P.h[h].gx[0]) continue;
                    const v4f o = {acc[h].x * f, acc[h].y * f, acc[h].z * f, acc[h].w * f};
                    __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(P.h[h].gx[l]) + row + v);     // a pure write stream
                }
            }
        }
    }
}

__global__ __launch_bounds__(kT) void rh_dw_kernel(const GP P) {
    __shared__ int s_ga[kRowChunk];
    __shared__ int s_wave[kT / 64];
    __shared__ float s_sum[kT / 64][4];
    extern __shared__ __align__(16) float s_dyn[];
    float* s_g = s_dyn;                                                 // [kRowChunk][4]  the rows' gradients of this workgroup's four channels
    float4* s_part = reinterpret_cast<float4*>(s_dyn + kRowChunk * 4);  // [4 waves][4 channels][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bid = blockIdx.x;
    int hi = 0;
    for (int i = 1; i < P.nheads; ++i) if (bid >= P.h[i].dw_blk0) hi = i;
    const HeadP& Hd = P.h[hi];
    int local = bid - Hd.dw_blk0;
    const int tap = local % 9; local /= 9;
    const int k = local / Hd.chunks, c0 = (local - k * Hd.chunks) * 4;
    const int ncc = Hd.no - c0 < 4 ? Hd.no - c0 : 4;                    // live channels of this chunk
    const int ky = tap / 3, kx = tap - 3 * ky;
    const int n = *P.num_pos;
    const float f = P.upstream ? *P.upstream : 1.0f;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int v0 = 0; v0 < P.cv; v0 += 64) {
        const int v = v0 + lane;
        const bool ok = v < P.cv;
        const int vc = ok ? v : P.cv - 1;
        float4 acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int r0 = 0; r0 < n; r0 += kRowChunk) {
            // ---- rows of anchor k in [r0, r0 + kRowChunk), ascending, into LDS (ballot compaction)
            int base = 0;
            for (int t0 = r0; t0 < r0 + kRowChunk && t0 < n; t0 += kT) {
                const int r = t0 + tid;
                int ga = 0; bool keep = false;
                if (r < n && r < r0 + kRowChunk) {
                    ga = P.pos[r];
                    const int a = ga - (ga / P.An) * P.An;
                    int l, cell, kk; decode_anchor(P, a, l, cell, kk);
                    keep = kk == k;
                }
                const unsigned long long m = __ballot(keep);
                if (lane == 0) s_wave[wv] = __popcll(m);
                __syncthreads();
                int off = base, tot = 0;
                for (int w = 0; w < kT / 64; ++w) { if (w < wv) off += s_wave[w]; tot += s_wave[w]; }
                if (keep) s_ga[off + __popcll(m & ((1ull << lane) - 1ull))] = ga;
                base += tot;
                __syncthreads();
            }
            const int cnt = base;
            for (int i = tid; i < cnt; i += kT) {
                const float* g = Hd.rows + grad_row(P, s_ga[i]) * Hd.no + c0;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                gv.x = g[0]; if (ncc > 1) gv.y = g[1]; if (ncc > 2) gv.z = g[2]; if (ncc > 3) gv.w = g[3];
                reinterpret_cast<float4*>(s_g)[i] = gv;
                if (v0 == 0 && tap == 0) { bsum[0] += gv.x; bsum[1] += gv.y; bsum[2] += gv.z; bsum[3] += gv.w; }
            }
            __syncthreads();
            if (Hd.gw) {
                // wave wv: rows wv, wv + 4, ... of the list, eight at a time (unconditional loads: out-of-image taps and the ragged end read
                // a valid address and are masked by a zero gradient)
                for (int i0 = wv; i0 < cnt; i0 += 32) {
                    float4 xv[8]; float4 gv[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const int i = i0 + 4 * u;
                        const bool live = i < cnt;
                        const int ga = s_ga[live ? i : i0], b = ga / P.An, a = ga - b * P.An;
                        int l, cell, kk; decode_anchor(P, a, l, cell, kk);
                        const Level L = P.lv[l];
                        const int y = cell / L.W, x = cell - y * L.W, yy = y + ky - 1, xx = x + kx - 1;
                        const bool in = yy >= 0 && yy < L.H && xx >= 0 && xx < L.W;
                        const size_t prow = (size_t)b * L.HW + (in ? yy * L.W + xx : cell);
                        xv[u] = reinterpret_cast<const float4*>(Hd.x[l])[prow * P.cv + vc];
                        gv[u] = (live && in) ? reinterpret_cast<const float4*>(s_g)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        const float gc[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
                        for (int c = 0; c < 4; ++c) {
                            acc[c].x += gc[c] * xv[u].x; acc[c].y += gc[c] * xv[u].y; acc[c].z += gc[c] * xv[u].z; acc[c].w += gc[c] * xv[u].w;
                        }
                    }
                }
            }
            __syncthreads();                                                    // the next chunk overwrites the lists
        }
        if (Hd.gw) {
#pragma unroll
            for (int c = 0; c < 4; ++c) s_part[(wv * 4 + c) * 64 + lane] = acc[c];
            __syncthreads();
            if (wv == 0 && ok) {
                for (int c = 0; c < ncc; ++c) {
                    float4 t = s_part[c * 64 + lane];
                    for (int w = 1; w < kT / 64; ++w) { const float4 u = s_part[(w * 4 + c) * 64 + lane]; t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w; }
                    const long long off = (long long)(k * Hd.no + c0 + c) * Hd.ws_o + ky * Hd.ws_ky + kx * Hd.ws_kx + (long long)(4 * v) * Hd.ws_i;
                    const float4 o = make_float4(t.x * f, t.y * f, t.z * f, t.w * f);
                    if (Hd.ws_i == 1) *reinterpret_cast<float4*>(Hd.gw + off) = o;
                    else { Hd.gw[off] = o.x; Hd.gw[off + Hd.ws_i] = o.y; Hd.gw[off + 2 * Hd.ws_i] = o.z; Hd.gw[off + 3 * Hd.ws_i] = o.w; }
                }
            }
            __syncthreads();
        }
    }
    if (Hd.gb && tap == 0) {                                                    // bias: the same rows without X (fixed tree + wave order)
#pragma unroll
        for (int c = 0; c < 4; ++c) bsum[c] = wave_sum(bsum[c]);
        if (lane == 0) { for (int c = 0; c < 4; ++c) s_sum[wv][c] = bsum[c]; }
        __syncthreads();
        if (tid < ncc) {
            float t = s_sum[0][tid];
            for (int w = 1; w < kT / 64; ++w) t += s_sum[w][tid];
            Hd.gb[k * Hd.no + c0 + tid] = t * f;
        }
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
constexpr size_t kPairLdsMax = 150 * 1024;
inline size_t pair_lds_bytes(int no, int cv) { const int slab = no < kSlabRows ? no : kSlabRows; return ((size_t)slab * cv * 4 + 16 * kSlabRows) * sizeof(float); }

bool valid(const odpl_retina_heads_desc_t* d, int n, const odpl_retina_head_t* h, long long* npix_img, long long* An) {
    if (!d || !h || n < 1 || n > kMaxHeads) return false;
    // A: the dx launch lists <= 96 (row, tap) contributions per pixel = 9 taps x A rows
    if (d->B < 1 || d->A < 1 || d->A > 10 || d->Cin < 4 || (d->Cin & 3) || d->nlevels < 1 || d->nlevels > ODPL_RETINA_LEVELS) return false;
    if (d->flags & ~ODPL_HEADS_SHARED_INPUT) return false;
    if ((d->flags & ODPL_HEADS_SHARED_INPUT) && n != 2) return false;
    if (!d->pos_anchor || !d->num_pos) return false;
    long long pix = 0;
    for (int l = 0; l < d->nlevels; ++l) { if (d->H[l] < 1 || d->W[l] < 1) return false; pix += (long long)d->H[l] * d->W[l]; }
    if (pix * d->B > 0x7fffffffLL || pix * d->A * d->B > 0x7fffffffLL || pix * d->B * d->Cin > 0x7fffffffLL * 4) return false;
    for (int i = 0; i < n; ++i) {
        if (h[i].no < 1 || h[i].no > 4096 || !h[i].w || !h[i].grad_rows) return false;
        if (pair_lds_bytes(h[i].no, d->Cin / 4) > kPairLdsMax) return false;          // the pair launch's weight slab in LDS: Cin <= 448 at no >= 80
        bool any_gx = false, all_gx = true;
        for (int l = 0; l < d->nlevels; ++l) { if (!h[i].x[l]) return false; any_gx |= h[i].grad_x[l] != nullptr; all_gx &= h[i].grad_x[l] != nullptr; }
        if (any_gx != all_gx) return false;
        if (h[i].w_stride[1] != 1 && h[i].w_stride[1] < 1) return false;
    }
    *npix_img = pix; *An = pix * d->A;
    if (d->rows_per_image != 0 && d->rows_per_image < *An) return false;
    if (d->flags & ODPL_HEADS_SHARED_INPUT)
        for (int l = 0; l < d->nlevels; ++l) if (h[0].x[l] != h[1].x[l] || h[0].grad_x[l] != h[1].grad_x[l]) return false;
    return true;
}
}  // namespace

extern "C" size_t odpl_retina_heads_backward_workspace_bytes(const odpl_retina_heads_desc_t* d, int32_t n, const odpl_retina_head_t* h) {
    long long pix, An;
    if (!valid(d, n, h, &pix, &An)) return 0;
    return 2 * align_up((size_t)pix * d->B * sizeof(int), 256) + (size_t)g_stage_rows * 9 * kMaxHeads * d->Cin * sizeof(float);
}

extern "C" int odpl_retina_heads_backward(const odpl_retina_heads_desc_t* d, int32_t n, const odpl_retina_head_t* heads,
                                          void* workspace, size_t workspace_bytes, void* stream_) {
    long long pix, An;
    if (!valid(d, n, heads, &pix, &An)) return ODPL_EINVAL;
    const size_t need = odpl_retina_heads_backward_workspace_bytes(d, n, heads);
    if (!workspace || workspace_bytes < need || ((uintptr_t)workspace & 15)) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    GP P = {};
    P.B = d->B; P.A = d->A; P.Cin = d->Cin; P.cv = d->Cin / 4; P.nlevels = d->nlevels; P.nheads = n;
    P.An = (int)An; P.npix_img = (int)pix; P.npix = (int)(pix * d->B);
    P.pos = d->pos_anchor; P.num_pos = d->num_pos; P.upstream = d->upstream;
    P.rows_an = d->rows_per_image ? d->rows_per_image : (int)An; P.shared_x = (d->flags & ODPL_HEADS_SHARED_INPUT) ? 1 : 0;
    int first = 0, pix0 = 0;
    for (int l = 0; l < d->nlevels; ++l) {
        P.lv[l].H = d->H[l]; P.lv[l].W = d->W[l]; P.lv[l].HW = d->H[l] * d->W[l]; P.lv[l].first = first; P.lv[l].pix0 = pix0;
        first += P.lv[l].HW * d->A; pix0 += P.lv[l].HW;
    }
    const size_t map_bytes = align_up((size_t)pix * d->B * sizeof(int), 256);
    P.map_first = (int*)workspace;
    P.map_cnt = (int*)((char*)workspace + map_bytes);
    P.stage = (float*)((char*)workspace + 2 * map_bytes);
    P.cap = g_stage_rows;
    int ndw = 0;
    bool want_x = false;
    for (int i = 0; i < n; ++i) {
        HeadP& H = P.h[i];
        for (int l = 0; l < d->nlevels; ++l) {
            H.x[l] = heads[i].x[l]; H.gx[l] = heads[i].grad_x[l];
            if (((uintptr_t)H.x[l] | (uintptr_t)H.gx[l]) & 15) return ODPL_EINVAL;
        }
        H.w = heads[i].w; H.gw = heads[i].grad_w; H.gb = heads[i].grad_b; H.rows = heads[i].grad_rows;
        H.ws_o = heads[i].w_stride[0]; H.ws_i = heads[i].w_stride[1]; H.ws_ky = heads[i].w_stride[2]; H.ws_kx = heads[i].w_stride[3];
        if (H.ws_i == 1 && ((((uintptr_t)H.w | (uintptr_t)H.gw) & 15) || (H.ws_o & 3) || (H.ws_ky & 3) || (H.ws_kx & 3))) return ODPL_EINVAL;
        H.no = heads[i].no; H.chunks = (H.no + 3) / 4;
        H.dw_blk0 = ndw; H.dw_nblk = (H.gw || H.gb) ? d->A * H.chunks * 9 : 0; ndw += H.dw_nblk;
        want_x |= H.gx[0] != nullptr;
    }
    ODPL_HIP_TRY(odpl::fill_bytes(P.map_cnt, 0, map_bytes, stream));
    hipLaunchKernelGGL(rh_map_kernel, dim3(256), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    if (want_x) {
        static size_t lds_allowed_dev[16] = {};              // hipFuncSetAttribute is per device
        int dev_id = 0;
        ODPL_HIP_TRY(hipGetDevice(&dev_id));
        if (dev_id < 0 || dev_id >= 16) return ODPL_EINVAL;
        size_t& lds_allowed = lds_allowed_dev[dev_id];
        if (lds_allowed == 0) lds_allowed = 64 * 1024;
        for (int i = 0; i < n; ++i) {
            if (!P.h[i].gx[0]) continue;
            const size_t lds_pair = pair_lds_bytes(P.h[i].no, P.cv);
            if (lds_pair > kPairLdsMax) return ODPL_EINVAL;          // valid() checks it
            if (lds_pair > lds_allowed) {
                ODPL_HIP_TRY(hipFuncSetAttribute((const void*)rh_pair_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_pair));
                lds_allowed = lds_pair;
            }
            P.pair_head = i;
            hipLaunchKernelGGL(rh_pair_kernel, dim3(d->A * 9 * kPairSlots), dim3(kT), lds_pair, stream, P);
            ODPL_HIP_TRY(hipGetLastError());
        }
        const int dx_waves = (P.npix + kPixPerWave - 1) / kPixPerWave;
        hipLaunchKernelGGL(rh_dx_kernel, dim3((unsigned)((dx_waves + kT / 64 - 1) / (kT / 64))), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    if (ndw > 0) {
        hipLaunchKernelGGL(rh_dw_kernel, dim3(ndw), dim3(kT), (size_t)kRowChunk * 4 * sizeof(float) + (size_t)16 * 64 * sizeof(float4), stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

// rows with staging slots in the workspace (default 4096; tests use a handful to exercise the in-place path of the dx launch)
extern "C" int odpl_retina_heads_tune(int32_t stage_rows) {
    if (stage_rows < 1 || stage_rows > (1 << 16)) return ODPL_EINVAL;
    g_stage_rows = stage_rows;
    return 0;
}
