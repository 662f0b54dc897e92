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
//   dx    one WAVE per level pixel: it looks at the <= 9 map entries of its 3 x 3 neighbourhood; nothing there -> a zero row (the dense
//         dX the previous layer's backward expects is written exactly once: an HBM write stream, 2 x 91 MB at config 4); otherwise the
//         contributions are summed in (tap, row, channel) order, lanes along Cin
//   dw    one workgroup per (head, anchor k, four output channels): the rows of anchor k are compacted into LDS in row order, then
//         summed four rows at a time (twelve X vectors in flight), taps split over the four waves, lanes along Cin
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;
constexpr int kMaxHeads = 2;
constexpr int kRowChunk = 2048;         // positive rows a dw workgroup stages in LDS at a time

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
    int B, A, Cin, cv, nlevels, nheads, An, npix_img, npix;      // npix_img = sum HW; npix = B * npix_img
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

__global__ __launch_bounds__(kT) void rh_dx_kernel(const GP P) {
    const int lane = threadIdx.x & 63;
    const int pix = blockIdx.x * (kT / 64) + (threadIdx.x >> 6);      // wave = one level pixel (wave-uniform from here on)
    if (pix >= P.npix) return;
    int l = 0;
    for (int i = 1; i < P.nlevels; ++i) if (pix >= P.B * P.lv[i].pix0) l = i;
    const Level L = P.lv[l];
    const int rel = pix - P.B * L.pix0, b = rel / L.HW, cell = rel - b * L.HW, y = cell / L.W, x = cell - y * L.W;
    const float f = P.upstream ? *P.upstream : 1.0f;
    // the 3 x 3 neighbourhood's map entries first: almost every pixel finds nothing and streams zeros
    int cnt[9], fst[9], any = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        const int ky = t / 3, kx = t - 3 * ky, yp = y - ky + 1, xp = x - kx + 1;      // the positive at (yp, xp) reaches (y, x) through tap (ky, kx)
        cnt[t] = 0; fst[t] = 0;
        if (yp >= 0 && yp < L.H && xp >= 0 && xp < L.W) {
            const int s = map_slot(P, l, b, yp * L.W + xp);
            cnt[t] = P.map_cnt[s];
            if (cnt[t]) fst[t] = P.map_first[s];
        }
        any |= cnt[t];
    }
    const size_t row = ((size_t)b * L.HW + cell) * P.cv;
    for (int v0 = 0; v0 < P.cv; v0 += 64) {
        const int v = v0 + lane;
        const bool ok = v < P.cv;
        float4 acc[kMaxHeads];
#pragma unroll
        for (int h = 0; h < kMaxHeads; ++h) acc[h] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (any && ok) {
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int ky = t / 3, kx = t - 3 * ky;
                for (int j = 0; j < cnt[t]; ++j) {
                    const int ga = __builtin_amdgcn_readfirstlane(P.pos[fst[t] + j]);
                    const int a = ga - (ga / P.An) * P.An;
                    const int k = (a - L.first) % P.A;
#pragma unroll
                    for (int h = 0; h < kMaxHeads; ++h) {
                        if (h >= P.nheads || !P.h[h].gx[0]) continue;
                        const HeadP& Hd = P.h[h];
                        const float* g = Hd.rows + (size_t)ga * Hd.no;
                        const long long base = (long long)k * Hd.no * Hd.ws_o + ky * Hd.ws_ky + kx * Hd.ws_kx + (long long)(4 * v) * Hd.ws_i;
                        int c = 0;
                        for (; c + 3 < Hd.no; c += 4) {                          // four weight rows in flight
                            const float4 w0 = ld_w4(Hd.w, base + (long long)c * Hd.ws_o, Hd.ws_i), w1 = ld_w4(Hd.w, base + (long long)(c + 1) * Hd.ws_o, Hd.ws_i);
                            const float4 w2 = ld_w4(Hd.w, base + (long long)(c + 2) * Hd.ws_o, Hd.ws_i), w3 = ld_w4(Hd.w, base + (long long)(c + 3) * Hd.ws_o, Hd.ws_i);
                            const float g0 = g[c], g1 = g[c + 1], g2 = g[c + 2], g3 = g[c + 3];
                            acc[h].x += g0 * w0.x; acc[h].y += g0 * w0.y; acc[h].z += g0 * w0.z; acc[h].w += g0 * w0.w;
                            acc[h].x += g1 * w1.x; acc[h].y += g1 * w1.y; acc[h].z += g1 * w1.z; acc[h].w += g1 * w1.w;
                            acc[h].x += g2 * w2.x; acc[h].y += g2 * w2.y; acc[h].z += g2 * w2.z; acc[h].w += g2 * w2.w;
                            acc[h].x += g3 * w3.x; acc[h].y += g3 * w3.y; acc[h].z += g3 * w3.z; acc[h].w += g3 * w3.w;
                        }
                        for (; c < Hd.no; ++c) {
                            const float4 w0 = ld_w4(Hd.w, base + (long long)c * Hd.ws_o, Hd.ws_i);
                            const float g0 = g[c];
                            acc[h].x += g0 * w0.x; acc[h].y += g0 * w0.y; acc[h].z += g0 * w0.z; acc[h].w += g0 * w0.w;
                        }
                    }
                }
            }
        }
        if (ok) {
#pragma unroll
            for (int h = 0; h < kMaxHeads; ++h) {
                if (h >= P.nheads || !P.h[h].gx[0]) continue;
                const v4f o = {acc[h].x * f, acc[h].y * f, acc[h].z * f, acc[h].w * f};
                __builtin_nontemporal_store(o, reinterpret_cast<v4f*>(P.h[h].gx[l]) + row + v);     // a pure write stream
            }
        }
    }
}

__global__ __launch_bounds__(kT) void rh_dw_kernel(const GP P) {
    __shared__ int s_ga[kRowChunk];
    __shared__ int s_wave[kT / 64];
    __shared__ float s_sum[kT / 64][4];
    extern __shared__ __align__(16) float s_g[];                        // [kRowChunk][4]  the rows' gradients of this workgroup's four channels
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, bid = blockIdx.x;
    int hi = 0;
    for (int i = 1; i < P.nheads; ++i) if (bid >= P.h[i].dw_blk0) hi = i;
    const HeadP& Hd = P.h[hi];
    const int local = bid - Hd.dw_blk0, k = local / Hd.chunks, c0 = (local - k * Hd.chunks) * 4;
    const int ncc = Hd.no - c0 < 4 ? Hd.no - c0 : 4;                    // live channels of this chunk
    const int n = *P.num_pos;
    const float f = P.upstream ? *P.upstream : 1.0f;
    // taps of this wave: wv, wv + 4, wv + 8 (< 9)
    int tap[3], ntap = 0;
    for (int t = wv; t < 9; t += 4) tap[ntap++] = t;
    float bsum[4] = {0.f, 0.f, 0.f, 0.f};
    for (int v0 = 0; v0 < P.cv; v0 += 64) {
        const int v = v0 + lane;
        const bool ok = v < P.cv;
        float4 acc[3][4];
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[t][c] = make_float4(0.f, 0.f, 0.f, 0.f);
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
                const float* g = Hd.rows + (size_t)s_ga[i] * Hd.no + c0;
                float4 gv = make_float4(0.f, 0.f, 0.f, 0.f);
                gv.x = g[0]; if (ncc > 1) gv.y = g[1]; if (ncc > 2) gv.z = g[2]; if (ncc > 3) gv.w = g[3];
                reinterpret_cast<float4*>(s_g)[i] = gv;
                if (v0 == 0) { bsum[0] += gv.x; bsum[1] += gv.y; bsum[2] += gv.z; bsum[3] += gv.w; }
            }
            __syncthreads();
            if (Hd.gw && ok) {
                for (int i0 = 0; i0 < cnt; i0 += 4) {                           // four rows at a time: up to twelve X vectors in flight
                    float4 xv[4][3];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int i = i0 + u < cnt ? i0 + u : cnt - 1;            // clamped (unconditional loads); masked below
                        const int ga = s_ga[i], b = ga / P.An, a = ga - b * P.An;
                        int l, cell, kk; decode_anchor(P, a, l, cell, kk);
                        const Level L = P.lv[l];
                        const int y = cell / L.W, x = cell - y * L.W;
#pragma unroll
                        for (int t = 0; t < 3; ++t) {
                            xv[u][t] = make_float4(0.f, 0.f, 0.f, 0.f);
                            if (t < ntap) {
                                const int ky = tap[t] / 3, kx = tap[t] - 3 * ky, yy = y + ky - 1, xx = x + kx - 1;
                                if (yy >= 0 && yy < L.H && xx >= 0 && xx < L.W)
                                    xv[u][t] = reinterpret_cast<const float4*>(Hd.x[l])[((size_t)b * L.HW + yy * L.W + xx) * P.cv + v];
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        if (i0 + u >= cnt) break;
                        const float4 gv = reinterpret_cast<const float4*>(s_g)[i0 + u];
                        const float gc[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
                        for (int t = 0; t < 3; ++t)
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                acc[t][c].x += gc[c] * xv[u][t].x; acc[t][c].y += gc[c] * xv[u][t].y;
                                acc[t][c].z += gc[c] * xv[u][t].z; acc[t][c].w += gc[c] * xv[u][t].w;
                            }
                    }
                }
            }
            __syncthreads();                                                    // the next chunk overwrites the lists
        }
        if (Hd.gw && ok) {
            for (int t = 0; t < ntap; ++t) {
                const int ky = tap[t] / 3, kx = tap[t] - 3 * ky;
                for (int c = 0; c < ncc; ++c) {
                    const long long off = (long long)(k * Hd.no + c0 + c) * Hd.ws_o + ky * Hd.ws_ky + kx * Hd.ws_kx + (long long)(4 * v) * Hd.ws_i;
                    const float4 o = make_float4(acc[t][c].x * f, acc[t][c].y * f, acc[t][c].z * f, acc[t][c].w * f);
                    if (Hd.ws_i == 1) *reinterpret_cast<float4*>(Hd.gw + off) = o;
                    else { Hd.gw[off] = o.x; Hd.gw[off + Hd.ws_i] = o.y; Hd.gw[off + 2 * Hd.ws_i] = o.z; Hd.gw[off + 3 * Hd.ws_i] = o.w; }
                }
            }
        }
    }
    if (Hd.gb) {                                                                // bias: the same rows without X (fixed tree + wave order)
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

bool valid(const odpl_retina_heads_desc_t* d, int n, const odpl_retina_head_t* h, long long* npix_img, long long* An) {
    if (!d || !h || n < 1 || n > kMaxHeads) return false;
    if (d->B < 1 || d->A < 1 || d->A > 64 || d->Cin < 4 || (d->Cin & 3) || d->nlevels < 1 || d->nlevels > ODPL_RETINA_LEVELS) return false;
    if (!d->pos_anchor || !d->num_pos) return false;
    long long pix = 0;
    for (int l = 0; l < d->nlevels; ++l) { if (d->H[l] < 1 || d->W[l] < 1) return false; pix += (long long)d->H[l] * d->W[l]; }
    if (pix * d->B > 0x7fffffffLL || pix * d->A * d->B > 0x7fffffffLL || pix * d->B * d->Cin > 0x7fffffffLL * 4) return false;
    for (int i = 0; i < n; ++i) {
        if (h[i].no < 1 || h[i].no > 4096 || !h[i].w || !h[i].grad_rows) return false;
        bool any_gx = false, all_gx = true;
        for (int l = 0; l < d->nlevels; ++l) { if (!h[i].x[l]) return false; any_gx |= h[i].grad_x[l] != nullptr; all_gx &= h[i].grad_x[l] != nullptr; }
        if (any_gx != all_gx) return false;
        if (h[i].w_stride[1] != 1 && h[i].w_stride[1] < 1) return false;
    }
    *npix_img = pix; *An = pix * d->A;
    return true;
}
}  // namespace

extern "C" size_t odpl_retina_heads_backward_workspace_bytes(const odpl_retina_heads_desc_t* d, int32_t n, const odpl_retina_head_t* h) {
    long long pix, An;
    if (!valid(d, n, h, &pix, &An)) return 0;
    return 2 * align_up((size_t)pix * d->B * sizeof(int), 256);
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
    int first = 0, pix0 = 0;
    for (int l = 0; l < d->nlevels; ++l) {
        P.lv[l].H = d->H[l]; P.lv[l].W = d->W[l]; P.lv[l].HW = d->H[l] * d->W[l]; P.lv[l].first = first; P.lv[l].pix0 = pix0;
        first += P.lv[l].HW * d->A; pix0 += P.lv[l].HW;
    }
    P.map_first = (int*)workspace;
    P.map_cnt = (int*)((char*)workspace + need / 2);
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
        H.dw_blk0 = ndw; H.dw_nblk = (H.gw || H.gb) ? d->A * H.chunks : 0; ndw += H.dw_nblk;
        want_x |= H.gx[0] != nullptr;
    }
    ODPL_HIP_TRY(odpl::fill_bytes(P.map_cnt, 0, need / 2, stream));
    hipLaunchKernelGGL(rh_map_kernel, dim3(256), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    if (want_x) {
        hipLaunchKernelGGL(rh_dx_kernel, dim3((unsigned)((P.npix + 3) / 4)), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    if (ndw > 0) {
        hipLaunchKernelGGL(rh_dw_kernel, dim3(ndw), dim3(kT), (size_t)kRowChunk * 4 * sizeof(float), stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}
