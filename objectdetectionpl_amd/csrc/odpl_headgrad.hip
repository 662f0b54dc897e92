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
// kernel the training step's loss-path roofline figure is quoted on.
//   stream  : workgroup = a tile of rows; lanes along Cin (16 B per lane, coalesced); gd of the tile staged in LDS; four
//             rows in flight per thread; non-temporal dX stores; per-tile dW / db partials (no float atomics)
//   finish  : fixed-order reduction of the partials -> dW / db of the dense channels; one workgroup per remaining output
//             channel (sum over the matched rows in row order); one workgroup per matched cell for the dX correction
//             (rows that share a cell are applied by the first of them, in row order)
// Bitwise reproducible run to run.
#include "odpl_common.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;

constexpr int kThreads = 256;
constexpr int kMaxA = ODPL_MAX_ANCHORS;
int g_rows_per_tile = 64;

struct HParams {
    const void* x; const float* w; void* gx; float* gw; float* gb;
    const float* gd; const float* gr; const int* row_cell; const float* upstream;
    float* part_w;        // [ntiles, A, Cin]
    float* part_b;        // [ntiles, A]
    int N, Cin, B, G2, A, no, dc, layout, n_rows, ntiles, rows_per_tile, tpr, rows_par;
    int fin_dense_blk0, fin_dense_nblk, fin_ch_blk0, fin_ch_nblk, fin_row_blk0;
};

template <bool BF>
__device__ __forceinline__ float4 ldx(const void* p, size_t quad) { return ld_quad<BF>(p, (long long)quad); }

// index of (row n, anchor a) in the dense part
__device__ __forceinline__ size_t gd_index(const HParams& P, int n, int a) {
    if (P.layout == ODPL_LAYOUT_PLANAR) { const int b = n / P.G2, s = n - b * P.G2; return ((size_t)b * P.A + a) * P.G2 + s; }
    return (size_t)n * P.A + a;
}
// (row n, anchor a) of a linear cell
__device__ __forceinline__ void cell_decode(const HParams& P, int cell, int& n, int& a) {
    if (P.layout == ODPL_LAYOUT_PLANAR) { const int plane = cell / P.G2, s = cell - plane * P.G2; a = plane % P.A; n = (plane / P.A) * P.G2 + s; }
    else { a = cell % P.A; n = cell / P.A; }
}

template <bool BF, int A>
__global__ __launch_bounds__(kThreads) void headgrad_stream_kernel(const HParams P) {
    extern __shared__ float lds[];                       // [rows_per_tile * A] gd of the tile, then [rows_par * A * tpr * 4] reduction
    const int tid = threadIdx.x, tile = blockIdx.x;
    const int n0 = tile * P.rows_per_tile;
    const int nrows = min(P.rows_per_tile, P.N - n0);
    const float f = P.upstream ? *P.upstream : 1.0f;
    float* s_g = lds;
    for (int i = tid; i < nrows * A; i += kThreads) {
        const int r = i / A, a = i - r * A;
        s_g[i] = P.gd[gd_index(P, n0 + r, a)] * f;
    }
    __syncthreads();
    const int tpr = P.tpr;                               // threads along Cin (power of two, <= 256)
    const int lane_c = tid & (tpr - 1), rl = tid / tpr;  // column slot, row lane
    const int rows_par = P.rows_par;                     // = kThreads / tpr row lanes
    const int cq = P.Cin / 4;                            // quads per row
    float* s_red = lds + P.rows_per_tile * A;
    float bsum[A];
#pragma unroll
    for (int a = 0; a < A; ++a) bsum[a] = 0.f;
    if (tid < A) { float s = 0.f; for (int r = 0; r < nrows; ++r) s += s_g[r * A + tid]; bsum[0] = s; }     // thread a: db partial of anchor a
    if (tid < A && P.part_b) P.part_b[(size_t)tile * A + tid] = bsum[0];

    for (int q0 = 0; q0 < cq; q0 += tpr) {               // column chunks (one for Cin <= 1024)
        const int q = q0 + lane_c;
        const bool col_ok = q < cq;
        float4 wv[A], acc[A];
#pragma unroll
        for (int a = 0; a < A; ++a) {
            acc[a] = make_float4(0.f, 0.f, 0.f, 0.f);
            wv[a] = col_ok ? reinterpret_cast<const float4*>(P.w + (size_t)(a * P.no + P.dc) * P.Cin)[q] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        if (col_ok) {
            int r = rl;
            for (; r + 3 * rows_par < nrows; r += 4 * rows_par) {        // four rows in flight
                float4 xv[4];
                if (P.gw) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) xv[u] = ldx<BF>(P.x, (size_t)(n0 + r + u * rows_par) * cq + q);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float* g = s_g + (r + u * rows_par) * A;
                    float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                    for (int a = 0; a < A; ++a) {
                        const float ga = g[a];
                        o.x += ga * wv[a].x; o.y += ga * wv[a].y; o.z += ga * wv[a].z; o.w += ga * wv[a].w;
                        if (P.gw) { acc[a].x += ga * xv[u].x; acc[a].y += ga * xv[u].y; acc[a].z += ga * xv[u].z; acc[a].w += ga * xv[u].w; }
                    }
                    if (P.gx) {
                        if (BF) st_quad<true>(P.gx, (long long)((size_t)(n0 + r + u * rows_par) * cq + q), o);
                        else __builtin_nontemporal_store(*reinterpret_cast<v4f*>(&o), reinterpret_cast<v4f*>(P.gx) + (size_t)(n0 + r + u * rows_par) * cq + q);
                    }
                }
            }
            for (; r < nrows; r += rows_par) {
                const float* g = s_g + r * A;
                float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
                if (P.gw) xv = ldx<BF>(P.x, (size_t)(n0 + r) * cq + q);
                float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    const float ga = g[a];
                    o.x += ga * wv[a].x; o.y += ga * wv[a].y; o.z += ga * wv[a].z; o.w += ga * wv[a].w;
                    if (P.gw) { acc[a].x += ga * xv.x; acc[a].y += ga * xv.y; acc[a].z += ga * xv.z; acc[a].w += ga * xv.w; }
                }
                if (P.gx) {
                    if (BF) st_quad<true>(P.gx, (long long)((size_t)(n0 + r) * cq + q), o);
                    else __builtin_nontemporal_store(*reinterpret_cast<v4f*>(&o), reinterpret_cast<v4f*>(P.gx) + (size_t)(n0 + r) * cq + q);
                }
            }
        }
        if (P.gw) {
            // combine the row lanes (fixed order) and write the tile's dW partial
            __syncthreads();
#pragma unroll
            for (int a = 0; a < A; ++a) reinterpret_cast<float4*>(s_red)[(rl * A + a) * tpr + lane_c] = acc[a];
            __syncthreads();
            if (rl == 0 && col_ok) {
#pragma unroll
                for (int a = 0; a < A; ++a) {
                    float4 s = reinterpret_cast<float4*>(s_red)[a * tpr + lane_c];
                    for (int k = 1; k < rows_par; ++k) {
                        const float4 v = reinterpret_cast<float4*>(s_red)[(k * A + a) * tpr + lane_c];
                        s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
                    }
                    reinterpret_cast<float4*>(P.part_w + ((size_t)tile * A + a) * P.Cin)[q] = s;
                }
            }
        }
    }
}

// finish: three kinds of workgroup in one launch
template <bool BF>
__global__ __launch_bounds__(kThreads) void headgrad_finish_kernel(const HParams P) {
    __shared__ float sh[kThreads / 64];
    const int tid = threadIdx.x, bid = blockIdx.x;
    const float f = P.upstream ? *P.upstream : 1.0f;
    const int cq = P.Cin / 4;
    if (bid >= P.fin_dense_blk0 && bid < P.fin_dense_blk0 + P.fin_dense_nblk) {
        // dense channels: dW[a*no+dc, :] = sum over tiles (ascending), db likewise
        const int a = (bid - P.fin_dense_blk0) % P.A, chunk = (bid - P.fin_dense_blk0) / P.A;
        const int q = chunk * kThreads + tid;
        const int ch = a * P.no + P.dc;
        if (P.gw && q < cq) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < P.ntiles; ++t) {
                const float4 v = reinterpret_cast<const float4*>(P.part_w + ((size_t)t * P.A + a) * P.Cin)[q];
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
            reinterpret_cast<float4*>(P.gw + (size_t)ch * P.Cin)[q] = s;
        }
        if (P.gb && chunk == 0) {
            float s = 0.f;
            for (int t = tid; t < P.ntiles; t += kThreads) s += P.part_b[(size_t)t * P.A + a];
            s = wave_sum(s);
            if ((tid & 63) == 0) sh[tid >> 6] = s;
            __syncthreads();
            if (tid == 0) { float v = sh[0]; for (int k = 1; k < kThreads / 64; ++k) v += sh[k]; P.gb[ch] = v; }
        }
        return;
    }
    if (bid >= P.fin_ch_blk0 && bid < P.fin_ch_blk0 + P.fin_ch_nblk) {
        // the other output channels: ch = a*no + c, c != dc; sum over the matched rows of anchor a, in row order
        const int idx = bid - P.fin_ch_blk0;
        const int a = idx / (P.no - 1);
        int c = idx - a * (P.no - 1); if (c >= P.dc) ++c;
        const int ch = a * P.no + c;
        float bsum = 0.f;
        for (int q0 = 0; q0 < cq || q0 == 0; q0 += kThreads) {
            const int q = q0 + tid;
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            for (int t = 0; t < P.n_rows; ++t) {
                const int cell = P.row_cell[t];
                if (cell < 0) continue;
                int n, ta; cell_decode(P, cell, n, ta);
                if (ta != a || n >= P.N) continue;
                const float g = P.gr[(size_t)t * P.no + c] * f;
                if (q0 == 0 && tid == 0) bsum += g;
                if (P.gw && q < cq) {
                    const float4 xv = ldx<BF>(P.x, (size_t)n * cq + q);
                    s.x += g * xv.x; s.y += g * xv.y; s.z += g * xv.z; s.w += g * xv.w;
                }
            }
            if (P.gw && q < cq) reinterpret_cast<float4*>(P.gw + (size_t)ch * P.Cin)[q] = s;
            if (cq == 0) break;
        }
        if (P.gb && tid == 0) P.gb[ch] = bsum;
        return;
    }
    // dX correction of the matched cells: one workgroup per row t; the first row of a cell-row n applies every row at n
    if (!P.gx) return;
    const int t0 = bid - P.fin_row_blk0;
    if (t0 < 0 || t0 >= P.n_rows) return;
    const int cell0 = P.row_cell[t0];
    if (cell0 < 0) return;
    int n0, a0; cell_decode(P, cell0, n0, a0);
    if (n0 >= P.N) return;
    for (int t = 0; t < t0; ++t) {                       // uniform: a lower row at the same n owns it
        const int cl = P.row_cell[t];
        if (cl < 0) continue;
        int n, a; cell_decode(P, cl, n, a);
        if (n == n0) return;
    }
    for (int q = tid; q < cq; q += kThreads) {
        float4 s = ldx<BF>(P.gx, (size_t)n0 * cq + q);
        for (int t = t0; t < P.n_rows; ++t) {
            const int cl = P.row_cell[t];
            if (cl < 0) continue;
            int n, a; cell_decode(P, cl, n, a);
            if (n != n0) continue;
            const float* g = P.gr + (size_t)t * P.no;
            for (int c = 0; c < P.no; ++c) {
                if (c == P.dc) continue;
                const float gv = g[c] * f;
                const float4 wv = reinterpret_cast<const float4*>(P.w + (size_t)(a * P.no + c) * P.Cin)[q];
                s.x += gv * wv.x; s.y += gv * wv.y; s.z += gv * wv.z; s.w += gv * wv.w;
            }
        }
        st_quad<BF>(P.gx, (long long)((size_t)n0 * cq + q), s);
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct HPlan { int ntiles, rows_per_tile, tpr, rows_par; size_t part_w, part_b, total; size_t lds; };

bool make_plan(const odpl_head_conv_t* h, const odpl_sparse_grad_t* g, HPlan* p) {
    if (!h || !g || !h->x || !h->w || h->Cin < 4 || h->Cin % 4 != 0) return false;
    if (g->B < 1 || g->G2 < 1 || g->A < 1 || g->A > kMaxA || g->no < 2 || g->dense_ch < 0 || g->dense_ch >= g->no || g->n_rows < 0) return false;
    if (g->layout != ODPL_LAYOUT_PLANAR && g->layout != ODPL_LAYOUT_INTERLEAVED) return false;
    if (!g->dense || (g->n_rows > 0 && (!g->rows || !g->row_cell))) return false;
    const long long N = (long long)g->B * g->G2;
    if (N * h->Cin > 0x7fffffffLL * 4LL) return false;
    int tpr = 1; while (tpr < h->Cin / 4 && tpr < kThreads) tpr <<= 1;
    p->tpr = tpr; p->rows_par = kThreads / tpr;
    p->rows_per_tile = g_rows_per_tile;
    p->ntiles = (int)((N + p->rows_per_tile - 1) / p->rows_per_tile);
    size_t off = 0;
    p->part_w = off; off += align_up((size_t)p->ntiles * g->A * h->Cin * sizeof(float), 256);
    p->part_b = off; off += align_up((size_t)p->ntiles * g->A * sizeof(float), 256);
    p->total = off;
    p->lds = ((size_t)p->rows_per_tile * g->A + (size_t)p->rows_par * g->A * tpr * 4) * sizeof(float);
    return p->lds <= 64 * 1024;
}

template <bool BF>
int launch(const HParams& P, const HPlan& pl, int A, int nfin, hipStream_t stream) {
    prof_begin(ODPL_PROF_HEADGRAD_STREAM, stream);
    switch (A) {
        case 1: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 1>), dim3(P.ntiles), dim3(kThreads), pl.lds, stream, P); break;
        case 2: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 2>), dim3(P.ntiles), dim3(kThreads), pl.lds, stream, P); break;
        case 3: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 3>), dim3(P.ntiles), dim3(kThreads), pl.lds, stream, P); break;
        case 4: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 4>), dim3(P.ntiles), dim3(kThreads), pl.lds, stream, P); break;
        case 5: hipLaunchKernelGGL((headgrad_stream_kernel<BF, 5>), dim3(P.ntiles), dim3(kThreads), pl.lds, stream, P); break;
        default: return ODPL_EINVAL;
    }
    prof_end(ODPL_PROF_HEADGRAD_STREAM, stream);
    ODPL_HIP_TRY(hipGetLastError());
    prof_begin(ODPL_PROF_HEADGRAD_FINISH, stream);
    hipLaunchKernelGGL((headgrad_finish_kernel<BF>), dim3(nfin), dim3(kThreads), 0, stream, P);
    prof_end(ODPL_PROF_HEADGRAD_FINISH, stream);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
}  // namespace

extern "C" size_t odpl_head_conv_backward_workspace_bytes(const odpl_head_conv_t* head, const odpl_sparse_grad_t* grad) {
    HPlan p;
    return make_plan(head, grad, &p) ? (p.total ? p.total : 256) : 0;
}

extern "C" int odpl_head_conv_backward(const odpl_head_conv_t* h, const odpl_sparse_grad_t* g, void* workspace,
                                       size_t workspace_bytes, void* stream_) {
    HPlan pl;
    if (!make_plan(h, g, &pl) || !workspace || workspace_bytes < pl.total || ((uintptr_t)workspace & 15) != 0) return ODPL_EINVAL;
    if (g->A > 5) return ODPL_EINVAL;
    if (((uintptr_t)h->x & 15) || ((uintptr_t)h->w & 15) || ((uintptr_t)h->grad_x & 15) || ((uintptr_t)h->grad_w & 15)) return ODPL_EINVAL;
    if (h->bf16 && h->Cin % 8 != 0) return ODPL_EINVAL;
    HParams P;
    P.x = h->x; P.w = h->w; P.gx = h->grad_x; P.gw = h->grad_w; P.gb = h->grad_b;
    P.gd = g->dense; P.gr = g->rows; P.row_cell = g->row_cell; P.upstream = h->upstream;
    char* ws = (char*)workspace;
    P.part_w = (float*)(ws + pl.part_w); P.part_b = (float*)(ws + pl.part_b);
    P.N = g->B * g->G2; P.Cin = h->Cin; P.B = g->B; P.G2 = g->G2; P.A = g->A; P.no = g->no; P.dc = g->dense_ch; P.layout = g->layout;
    P.n_rows = g->n_rows; P.ntiles = pl.ntiles; P.rows_per_tile = pl.rows_per_tile; P.tpr = pl.tpr; P.rows_par = pl.rows_par;
    const int cq = h->Cin / 4;
    int nblk = 0;
    P.fin_dense_blk0 = nblk; P.fin_dense_nblk = g->A * ((cq + kThreads - 1) / kThreads); nblk += P.fin_dense_nblk;
    P.fin_ch_blk0 = nblk; P.fin_ch_nblk = (h->grad_w || h->grad_b) ? g->A * (g->no - 1) : 0; nblk += P.fin_ch_nblk;
    P.fin_row_blk0 = nblk; nblk += h->grad_x ? g->n_rows : 0;
    hipStream_t stream = (hipStream_t)stream_;
    return h->bf16 ? launch<true>(P, pl, g->A, nblk, stream) : launch<false>(P, pl, g->A, nblk, stream);
}

extern "C" int odpl_head_conv_tune(int32_t rows_per_tile) {
    if (rows_per_tile < 4 || rows_per_tile > 1024) return ODPL_EINVAL;
    g_rows_per_tile = rows_per_tile;
    return 0;
}
