// Forward of a 3x3 (stride 1, pad 1) convolution on the pixels a later reader will actually look at.
//
// RetinaNetLoss (losses.py:405-472) reads the POSITIVE anchor rows of (loc_preds, cls_preds) and nothing else: SmoothL1 runs on
// positives (:458-461) and the focal one-hot has no background column (:336-339), so a background row's term is 0 * log p.  At BASELINE
// config 4 that is 863 of 802 800 rows, on ~600 of 89 200 level pixels.  In a TRAINING step (step.py:14-25: forward, criterion, nothing
// else looks at the predictions) the two final head convolutions are therefore only needed on the positive pixels S0, the tower layer
// feeding them within one pixel of S0, the layer before within two ... (model/RetinaNet.py:93-99: four Conv2d(256, 256, 3, 1, 1) + ReLU
// per head).  Target assignment needs the anchors and the boxes only, so it can run BEFORE the heads:
//     odpl_retina_loss(ODPL_RETINA_ASSIGN_ONLY) -> positive rows
//     odpl_pixel_support                          -> per level, per reach j = 0 .. D-1: the pixels within Chebyshev distance j of S0
//     odpl_sparse_conv3x3_forward                 -> y[p, :] = act(b + sum_tap W_tap . x[p + d(tap), :]) for p in one of those lists
// The product is a gathered GEMM on the fp32 matrix cores (v_mfma_f32_32x32x2_f32; 64 x 64 tiles, K step 32, the rows of x gathered one
// tap at a time, padded LDS images, the next K tile prefetched into registers) -- the tiling of csrc/odpl_sparseconv.hip, whose backward
// kernels then walk the same pixel sets.  Values on the listed pixels are those of the dense convolution (same products, fp32
// accumulation in tap-major order); pixels off the list are left as the caller initialised them.  The lists are work queues: their
// order is not defined (wave-aggregated integer atomics) and no result depends on it.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kT = 256;
constexpr int BK = 32;
constexpr int kMaxLevels = 8;
constexpr int kMaxDepth = 8;

struct SupP {
    const int* pos_anchor; const int* num_pos;
    int B, An, A, n_levels, depth, level_mask, total;   // total = pixels of the wanted levels
    int H[kMaxLevels], W[kMaxLevels];
    int first[kMaxLevels + 1];                        // first anchor row of a level
    int pbase[kMaxLevels + 1];                        // first pixel (over all levels, batch inside the level) of a level
    int n_wanted, wlevel[kMaxLevels];                 // the levels lists are wanted for
    int wbase[kMaxLevels + 1];                        // first thread of the k-th wanted level (each range padded to whole waves)
    unsigned char* flags;                             // [pbase[n_levels]] 1 = a positive anchor sits on this pixel
    int* lists; int* counts;
};

__global__ __launch_bounds__(kT) void sup_mark_kernel(const SupP P) {
    const int n = P.num_pos[0];
    for (int r = blockIdx.x * kT + threadIdx.x; r < n; r += gridDim.x * kT) {
        const int ga = P.pos_anchor[r], b = ga / P.An, a = ga - b * P.An;
        int l = 0;
        while (l + 1 < P.n_levels && a >= P.first[l + 1]) ++l;
        const int pix = (a - P.first[l]) / P.A;
        P.flags[P.pbase[l] + b * P.H[l] * P.W[l] + pix] = 1;
    }
}

// thread = pixel of a wanted level (each level's range padded to whole waves, so a wave sits on one level): Chebyshev distance to
// the nearest marked pixel (up to depth - 1), then one wave-aggregated list append per reach
__global__ __launch_bounds__(kT) void sup_list_kernel(const SupP P) {
    const int t = blockIdx.x * kT + threadIdx.x, lane = threadIdx.x & 63;
    int k = 0;
    while (k + 1 < P.n_wanted && t >= P.wbase[k + 1]) ++k;
    const int l = P.wlevel[k], q = t - P.wbase[k];
    const int H = P.H[l], W = P.W[l], hw = H * W;
    const bool live = q < P.B * hw;
    int dmin = kMaxDepth + 1;
    if (live) {
        const int b = q / hw, rem = q - b * hw, y = rem / W, x = rem - y * W;
        const unsigned char* f = P.flags + P.pbase[l] + b * hw;
        const int R = P.depth - 1;
        for (int dy = -R; dy <= R; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -R; dx <= R; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                if (f[yy * W + xx]) {
                    const int ay = dy < 0 ? -dy : dy, ax = dx < 0 ? -dx : dx, dd = ay > ax ? ay : ax;
                    dmin = dd < dmin ? dd : dmin;
                }
            }
        }
    }
    for (int j = 0; j < P.depth; ++j) {
        const bool in = live && dmin <= j;
        const unsigned long long m = __ballot(in);
        if (m == 0) continue;                                            // wave-uniform
        const int lead = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == lead) base = atomicAdd(P.counts + l * P.depth + j, __popcll(m));
        base = __shfl(base, lead, 64);
        if (in) P.lists[(size_t)P.pbase[l] * P.depth + (size_t)j * P.B * hw + base + __popcll(m & ((1ull << lane) - 1ull))] = q;
    }
}

struct FwdP {
    const float* x; const float* w; const float* bias; float* y;
    long long ws_o, ws_i, ws_ky, ws_kx, ybs, yps;
    const int* list; const int* count;
    int B, H, W, Cin, Cout, act;
    int ks, pad, taps;                   // 3 x 3 (pad 1) or 1 x 1 (pad 0)
    float* part; long long part_floats;  // split-K partials [S][tiles_m * 64][tiles_n * 64] (null: never split)
};

// A short list makes a grid of a few dozen workgroups, each walking all 9 * Cin / 32 K tiles at the matrix pipe's own pace (16 MFMAs of
// 64 cycles per tile and wave): the launch takes as long as ONE tile's K loop however idle the chip is.  K is therefore ALWAYS summed as
// R = 8 consecutive ranges (range sums added left to right), and a short list gives each range to its own workgroup; a second launch
// adds the R partials in the same order and applies the epilogue.  The result does not depend on which way a launch went -- bit for bit.
__device__ __forceinline__ int k_ranges(int nk) { return nk < 8 ? nk : 8; }
__device__ __forceinline__ int k_bound(int r, int nk, int R) { return (int)((long long)r * nk / R); }
__device__ __forceinline__ bool split_ranges(int ntiles, int tiles_m, int tiles_n, int R, const float* part, long long part_floats) {
    return part && R > 1 && ntiles * R <= 1024 && (long long)tiles_m * 64 * tiles_n * 64 * R <= part_floats;
}

// y rows of the listed pixels: [n, 9 * Cin] (rows of x gathered one tap at a time) x [9 * Cin, Cout]
__global__ __launch_bounds__(kT) void sf_conv_kernel(const FwdP P) {
    constexpr int BM = 64, BN = 64, LD = BK + 1;
    constexpr int D = 4;                                                 // K tiles in flight per workgroup: these grids are a few dozen
                                                                         // workgroups walking 72 K tiles each -- latency, not bandwidth
    __shared__ float As[2][BM * LD];
    __shared__ float Bs[2][BN * LD];
    __shared__ int s_b[BM], s_y[BM], s_x[BM];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wm = (wv >> 1) * 32, wn = (wv & 1) * 32, lr = lane & 31, lk = lane >> 5;
    int n = P.count[0];
    const int cap = P.B * P.H * P.W;
    if (n > cap) n = cap;
    const int tiles_n = (P.Cout + BN - 1) / BN, tiles_m = (n + BM - 1) / BM, ntiles = tiles_m * tiles_n;
    const int hw = P.H * P.W, kpt = P.Cin / BK;                          // K steps per tap
    const int nk_all = P.taps * kpt, R = k_ranges(nk_all);
    const int S = split_ranges(ntiles, tiles_m, tiles_n, R, P.part, P.part_floats) ? R : 1;
    for (int work = blockIdx.x; work < ntiles * S; work += gridDim.x) {
        const int tile = work / S, sp = work - tile * S;
        const int k_begin = S == 1 ? 0 : k_bound(sp, nk_all, R), nk = S == 1 ? nk_all : k_bound(sp + 1, nk_all, R) - k_begin;
        const int tm = tile / tiles_n, tn = tile - tm * tiles_n, bn = tn * BN;
        __syncthreads();
        if (tid < BM) {
            int r = tm * BM + tid; if (r > n - 1) r = n - 1;             // clamped: the surplus rows are computed and not stored
            const int q = P.list[r], b = q / hw, rem = q - b * hw, y = rem / P.W;
            s_b[tid] = b; s_y[tid] = y; s_x[tid] = rem - y * P.W;
        }
        __syncthreads();
        f32x16 acc, total;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[r] = 0.f; total[r] = 0.f; }
        int cur = 0, next_bound = S == 1 ? k_bound(1, nk_all, R) : nk;       // unsplit: fold the range sums in order as they complete
        float4 ra[D][2], rb[D][2];
        auto load_tiles = [&](int kl, float4 (&xa)[2], float4 (&xb)[2]) {          // kl: K tile within this range
            const int kt = k_begin + kl;
            const int t = kt / kpt, ci0 = (kt - t * kpt) * BK, ky = t / P.ks, kx = t - P.ks * ky;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, row = f >> 3, kq = f & 7;
                const int yy = s_y[row] + (ky - P.pad), xx = s_x[row] + (kx - P.pad);        // y[p] takes x[p + d(tap)]
                const bool in = yy >= 0 && yy < P.H && xx >= 0 && xx < P.W;
                const size_t src = ((size_t)s_b[row] * hw + (in ? yy * P.W + xx : 0)) * P.Cin + ci0 + 4 * kq;
                const float4 v = *reinterpret_cast<const float4*>(P.x + src);
                xa[i] = in ? v : make_float4(0.f, 0.f, 0.f, 0.f);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, nr = f >> 3, kq = f & 7;                         // W[bn + nr][ci0 + 4 kq ..][ky][kx]
                int co = bn + nr; if (co > P.Cout - 1) co = P.Cout - 1;                      // surplus columns: computed, not stored
                const long long off = (long long)co * P.ws_o + ky * P.ws_ky + kx * P.ws_kx + (long long)(ci0 + 4 * kq) * P.ws_i;
                xb[i] = P.ws_i == 1 ? *reinterpret_cast<const float4*>(P.w + off)
                                    : make_float4(P.w[off], P.w[off + P.ws_i], P.w[off + 2 * P.ws_i], P.w[off + 3 * P.ws_i]);
            }
        };
        auto store_tiles = [&](int buf, const float4 (&xa)[2], const float4 (&xb)[2]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int f = tid + kT * i, row = f >> 3, kq = f & 7;
                float* d = As[buf] + row * LD + 4 * kq;
                d[0] = xa[i].x; d[1] = xa[i].y; d[2] = xa[i].z; d[3] = xa[i].w;
                float* e = Bs[buf] + row * LD + 4 * kq;
                e[0] = xb[i].x; e[1] = xb[i].y; e[2] = xb[i].z; e[3] = xb[i].w;
            }
        };
        // K tile t travels: global -> register set t % D (requested D - 1 iterations before it is needed) -> LDS image (t & 1) -> MFMA
#pragma unroll
        for (int j = 0; j < D; ++j)
            if (j < nk) load_tiles(j, ra[j], rb[j]);
        store_tiles(0, ra[0], rb[0]);
        __syncthreads();
        for (int kt0 = 0; kt0 < nk; kt0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int kt = kt0 + j;
                if (kt < nk) {
                    if (kt + 1 < nk) store_tiles((j + 1) & 1, ra[(j + 1) % D], rb[(j + 1) % D]);   // (kt0 is a multiple of D, D even)
                    if (kt + D < nk) load_tiles(kt + D, ra[j], rb[j]);                             // set j was stored one iteration ago
                    const float* A = As[j & 1];
                    const float* Bm = Bs[j & 1];
#pragma unroll
                    for (int kk = 0; kk < BK; kk += 2) {
                        const float a = A[(wm + lr) * LD + kk + lk], b = Bm[(wn + lr) * LD + kk + lk];
                        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
                    }
                    if (S == 1 && kt + 1 == next_bound) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) { total[r] += acc[r]; acc[r] = 0.f; }
                        ++cur; next_bound = k_bound(cur + 1, nk_all, R);
                    }
                    __syncthreads();
                }
            }
        }
        if (S == 1) acc = total;
        const int co = bn + wn + lr;
        if (S > 1) {                                                     // partial sums of this K range; sf_finish_kernel does the rest
            float* o = P.part + ((size_t)sp * tiles_m * BM + (size_t)tm * BM) * ((size_t)tiles_n * BN) + bn + wn + lr;
#pragma unroll
            for (int r = 0; r < 16; ++r) o[(size_t)(wm + (r & 3) + 8 * (r >> 2) + 4 * lk) * ((size_t)tiles_n * BN)] = acc[r];
            continue;
        }
        const float bv = (P.bias && co < P.Cout) ? P.bias[co] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm + (r & 3) + 8 * (r >> 2) + 4 * lk;
            if (tm * BM + row < n && co < P.Cout) {
                float v = acc[r] + bv;
                if (P.act == ODPL_ACT_RELU) v = v > 0.f ? v : 0.f;
                else if (P.act == ODPL_ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
                P.y[(long long)s_b[row] * P.ybs + (long long)(s_y[row] * P.W + s_x[row]) * P.yps + co] = v;
            }
        }
    }
}

// the split launches' second half: y[q, co] = act(bias + sum over the S ranges, in range order); returns at once when S = 1
__global__ __launch_bounds__(kT) void sf_finish_kernel(const FwdP P) {
    int n = P.count[0];
    const int cap = P.B * P.H * P.W;
    if (n > cap) n = cap;
    const int tiles_n = (P.Cout + 63) / 64, tiles_m = (n + 63) / 64, kpt = P.Cin / BK;
    const int R = k_ranges(P.taps * kpt);
    if (!split_ranges(tiles_m * tiles_n, tiles_m, tiles_n, R, P.part, P.part_floats)) return;
    const int S = R;
    const int hw = P.H * P.W, ld = tiles_n * 64;
    const size_t plane = (size_t)tiles_m * 64 * ld;
    const long long total = (long long)n * P.Cout;
    for (long long i = (long long)blockIdx.x * kT + threadIdx.x; i < total; i += (long long)gridDim.x * kT) {
        const int r = (int)(i / P.Cout), co = (int)(i - (long long)r * P.Cout);
        float v = 0.f;                                                   // 0 + r0 + r1 + ...: the order of the unsplit launch
        for (int s2 = 0; s2 < S; ++s2) v += P.part[s2 * plane + (size_t)r * ld + co];
        if (P.bias) v += P.bias[co];
        if (P.act == ODPL_ACT_RELU) v = v > 0.f ? v : 0.f;
        else if (P.act == ODPL_ACT_LEAKY) v = v > 0.f ? v : 0.1f * v;
        const int q = P.list[r], b = q / hw, rem = q - b * hw;
        P.y[(long long)b * P.ybs + (long long)rem * P.yps + co] = v;
    }
}

// m[q, :] = lat[q, :] + bilinear(coarse)[q, :] for the listed pixels q of the fine map (FPN top-down merge, RetinaNetbone.py:57
// `F.upsample(x, size, mode='bilinear') + y`, align_corners = False): one wave per pixel, lanes across channel quads
struct UpP {
    const float* coarse; const float* lat; float* out;
    const int* list; const int* count;
    int B, h, w, H, W, C;
    float sy, sx;                        // h / H, w / W (ATen's area_pixel_compute_scale)
};

__global__ __launch_bounds__(kT) void sf_upadd_kernel(const UpP P) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, c4n = P.C >> 2;
    int n = P.count[0];
    const int cap = P.B * P.H * P.W;
    if (n > cap) n = cap;
    for (int r = blockIdx.x * (kT / 64) + wv; r < n; r += gridDim.x * (kT / 64)) {
        const int q = P.list[r], HW = P.H * P.W, b = q / HW, rem = q - b * HW, y = rem / P.W, x = rem - y * P.W;
        float fy = P.sy * ((float)y + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;          // ATen: area_pixel_compute_source_index, cubic = false
        float fx = P.sx * ((float)x + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < P.h - 1 ? 1 : 0), x1 = x0 + (x0 < P.w - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
        const float4* c00 = reinterpret_cast<const float4*>(P.coarse + ((size_t)(b * P.h + y0) * P.w + x0) * P.C);
        const float4* c01 = reinterpret_cast<const float4*>(P.coarse + ((size_t)(b * P.h + y0) * P.w + x1) * P.C);
        const float4* c10 = reinterpret_cast<const float4*>(P.coarse + ((size_t)(b * P.h + y1) * P.w + x0) * P.C);
        const float4* c11 = reinterpret_cast<const float4*>(P.coarse + ((size_t)(b * P.h + y1) * P.w + x1) * P.C);
        const float4* l = reinterpret_cast<const float4*>(P.lat + (size_t)q * P.C);
        float4* o = reinterpret_cast<float4*>(P.out + (size_t)q * P.C);
        for (int c = lane; c < c4n; c += 64) {
            const float4 a = c00[c], bq = c01[c], cq = c10[c], dq = c11[c], lv = l[c];
            float4 v;                                                    // ATen's order: h0 * (w0 * v00 + w1 * v01) + h1 * (w0 * v10 + w1 * v11)
            v.x = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * cq.x + lx1 * dq.x) + lv.x;
            v.y = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * cq.y + lx1 * dq.y) + lv.y;
            v.z = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * cq.z + lx1 * dq.z) + lv.z;
            v.w = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * cq.w + lx1 * dq.w) + lv.w;
            o[c] = v;
        }
    }
}

bool fill_sup(const odpl_pixel_support_t* d, SupP* P) {
    if (!d || d->B < 1 || d->An < 1 || d->A < 1 || d->n_levels < 1 || d->n_levels > kMaxLevels || d->depth < 1 || d->depth > kMaxDepth) return false;
    if (!d->pos_anchor || !d->num_pos || !d->lists || !d->counts || d->level_mask == 0 || (d->level_mask >> d->n_levels)) return false;
    long long first = 0, pb = 0, wb = 0;
    int seen = 0;
    for (int l = 0; l < d->n_levels; ++l) {
        if (d->H[l] < 1 || d->W[l] < 1) return false;
        P->H[l] = d->H[l]; P->W[l] = d->W[l];
        P->first[l] = (int)first; P->pbase[l] = (int)pb;
        if ((d->level_mask >> l) & 1) { P->wlevel[seen] = l; P->wbase[seen++] = (int)wb; wb += ((long long)d->B * d->H[l] * d->W[l] + 63) / 64 * 64; }
        first += (long long)d->A * d->H[l] * d->W[l];
        pb += (long long)d->B * d->H[l] * d->W[l];
        if (pb * d->depth > 0x7fffffffLL) return false;
    }
    if (first != d->An) return false;                                    // the level maps must tile the anchor table
    P->first[d->n_levels] = (int)first; P->pbase[d->n_levels] = (int)pb;
    P->n_wanted = seen;
    for (int k = seen; k <= kMaxLevels; ++k) P->wbase[k] = 0x7fffffff;
    for (int k = seen; k < kMaxLevels; ++k) P->wlevel[k] = 0;
    P->pos_anchor = d->pos_anchor; P->num_pos = d->num_pos;
    P->B = d->B; P->An = d->An; P->A = d->A; P->n_levels = d->n_levels; P->depth = d->depth; P->level_mask = d->level_mask; P->total = (int)wb;
    P->lists = d->lists; P->counts = d->counts;
    return true;
}
}  // namespace

extern "C" size_t odpl_pixel_support_workspace_bytes(const odpl_pixel_support_t* d) {
    SupP P;
    return fill_sup(d, &P) ? (((size_t)P.pbase[P.n_levels] + 255) / 256 * 256) : 0;
}

extern "C" int odpl_pixel_support(const odpl_pixel_support_t* d, void* workspace, size_t workspace_bytes, void* stream_) {
    SupP P;
    if (!fill_sup(d, &P) || !workspace || workspace_bytes < (size_t)P.pbase[P.n_levels]) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    P.flags = (unsigned char*)workspace;
    ODPL_HIP_TRY(hipMemsetAsync(P.flags, 0, (size_t)P.pbase[P.n_levels], stream));
    ODPL_HIP_TRY(hipMemsetAsync(P.counts, 0, sizeof(int) * (size_t)P.n_levels * P.depth, stream));
    hipLaunchKernelGGL(sup_mark_kernel, dim3(64), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(sup_list_kernel, dim3((P.total + kT - 1) / kT), dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int odpl_sparse_conv3x3_forward(const odpl_sparse_conv3x3_fwd_t* d, void* workspace, size_t workspace_bytes, void* stream_) {
    if (!d || d->B < 1 || d->H < 1 || d->W < 1 || d->Cin < BK || (d->Cin % BK) || d->Cout < 1) return ODPL_EINVAL;
    if (!d->x || !d->w || !d->y || !d->list || !d->count) return ODPL_EINVAL;
    if (d->act != ODPL_ACT_LINEAR && d->act != ODPL_ACT_RELU && d->act != ODPL_ACT_LEAKY) return ODPL_EINVAL;
    if (d->ksize != 0 && d->ksize != 1 && d->ksize != 3) return ODPL_EINVAL;
    const long long Pn = (long long)d->B * d->H * d->W;
    if (Pn * d->Cin > 0x7fffffffLL * 4 || Pn > 0x7fffffffLL - 256) return ODPL_EINVAL;
    if (((uintptr_t)d->x) & 15) return ODPL_EINVAL;
    if (d->w_stride[1] == 1 && ((((uintptr_t)d->w) & 15) || (d->w_stride[0] & 3) ||
                                (d->ksize != 1 && ((d->w_stride[2] & 3) || (d->w_stride[3] & 3))))) return ODPL_EINVAL;   // (a 1 x 1 kernel never steps along ky / kx)
    FwdP P;
    P.x = d->x; P.w = d->w; P.bias = d->bias; P.y = d->y;
    P.ws_o = d->w_stride[0]; P.ws_i = d->w_stride[1]; P.ws_ky = d->w_stride[2]; P.ws_kx = d->w_stride[3];
    P.ybs = d->y_batch_stride; P.yps = d->y_pixel_stride;
    P.list = d->list; P.count = d->count;
    P.B = d->B; P.H = d->H; P.W = d->W; P.Cin = d->Cin; P.Cout = d->Cout; P.act = d->act;
    P.ks = d->ksize == 1 ? 1 : 3; P.pad = P.ks / 2; P.taps = P.ks * P.ks;
    const bool split = workspace && workspace_bytes >= (1u << 20) && !(((uintptr_t)workspace) & 15);   // optional: short lists cut K over workgroups
    P.part = split ? (float*)workspace : nullptr; P.part_floats = split ? (long long)(workspace_bytes / sizeof(float)) : 0;
    const long long worst = (Pn + 63) / 64 * ((d->Cout + 63) / 64);
    hipLaunchKernelGGL(sf_conv_kernel, dim3((unsigned)(worst < 2048 ? worst : 2048)), dim3(kT), 0, (hipStream_t)stream_, P);
    ODPL_HIP_TRY(hipGetLastError());
    if (split) {
        hipLaunchKernelGGL(sf_finish_kernel, dim3(512), dim3(kT), 0, (hipStream_t)stream_, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int odpl_sparse_upsample_add(const float* coarse, const float* lateral, float* out, const int32_t* list, const int32_t* count,
                                        int32_t B, int32_t C, int32_t h, int32_t w, int32_t H, int32_t W, void* stream_) {
    if (!coarse || !lateral || !out || !list || !count || B < 1 || C < 4 || (C & 3) || h < 1 || w < 1 || H < 1 || W < 1) return ODPL_EINVAL;
    if ((((uintptr_t)coarse | (uintptr_t)lateral | (uintptr_t)out) & 15) || (long long)B * H * W > 0x7fffffffLL) return ODPL_EINVAL;
    UpP P;
    P.coarse = coarse; P.lat = lateral; P.out = out; P.list = list; P.count = count;
    P.B = B; P.h = h; P.w = w; P.H = H; P.W = W; P.C = C;
    P.sy = (float)h / (float)H; P.sx = (float)w / (float)W;
    const long long waves = (long long)B * H * W;
    const long long blocks = (waves + kT / 64 - 1) / (kT / 64);
    hipLaunchKernelGGL(sf_upadd_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096)), dim3(kT), 0, (hipStream_t)stream_, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
