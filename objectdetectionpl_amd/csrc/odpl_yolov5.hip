// YOLOv5 loss for MI355X (gfx950): build_targets_v5 (accuracy.py:472-521) + MultiScaleRegionLoss_v5.forward
// (losses.py:98-156) with FocalLoss (losses.py:37-64) and bbox_iou_v5 in IoU/GIoU/DIoU/CIoU mode
// (accuracy.py:71-114), forward AND d loss/d p.
//
// Layout: p[i] is the [B, na, G, G, no] tensor Yolo_Layers.forward hands the criterion (YOLOV5.py:93-97) in either memory
// format -- both use the cell index (b*na + a)*G*G + s:
//   interleaved  contiguous [B,na,G,G,no]: element (cell, c) at cell*no + c            (the reference's permute().contiguous())
//   planar       the Detect convolution's own NCHW output viewed as [B,na,no,G,G]:     element at ((cell/G2)*no + c)*G2 + cell%G2
//                -- the permute is a VIEW, no copy is made, and the objectness logits are contiguous planes
// Heads and their gradients may be bf16 (desc.bf16: BASELINE config 5 trains under bf16 autocast): loads widen to fp32,
// every sum stays fp32, gradients are rounded once.
//
// Like the v2/v3/v4 region loss the gradient is STRUCTURED: one dense value per cell (the objectness logit) + `no` values on
// each matched row.  forward leaves it compact (gobj[cells], row_grad[rows, no], row_cell[rows]):
//   assign (1 WG / layer)  ratio test < 4, 'rect4' neighbour expansion, rows in the reference's order
//                          [base | x-1 | y-1 | x+1 | y+1] by ballot/popcount compaction; owner map
//                          atomicMax(row) => "last row wins" for tobj (and marks each cell's leader row)
//   rows   (1 wave / row)  the cell's no logits, box decode, IoU-family value AND its gradient by forward-mode
//                          differentiation (4 tangents), focal-BCE over the classes -> row_grad
//   obj    (thread = cell) objectness focal-BCE(x, tobj) value + gobj[cell]; owner map reset.  planar: the logits are
//                          contiguous planes (coalesced, no over-fetch); interleaved: one request per cell
//   finalize (1 WG)        fixed-order sums -> lbox*0.05 + lobj*1.0 + lcls*0.58
// backward: a 1x1-convolution producer consumes the compact form directly (odpl_headgrad.hip, no dense tensor); any other
// producer gets the dense d loss/d p:
//   expand (chunks)        pure 16-B non-temporal store stream over the gradient tensor (zeros + gobj in the objectness
//                          slots), scaled by the upstream scalar           [HBM-write bound]
//   cells  (1 wave / row)  leader rows add the row gradients of all rows that share their cell, ascending row
//                          order (index_put accumulate semantics, deterministic), over the zeros
#include <hip/hip_bf16.h>
#include "odpl_common.hpp"
#include "odpl_yolo_dev.hpp"
#include "odpl_profile.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kAT = 256;              // assign threads
constexpr int kDT = 256;              // dense threads
constexpr int kChunkF4 = 4096;        // float4 per dense workgroup

struct Layer {
    const void* p; void* grad;        // fp32 or bf16 (VParams.bf16): loads widen to fp32, sums stay fp32, the gradient is rounded once
    int G, G2, cells, layout;         // cells = B*na*G*G; layout: ODPL_LAYOUT_*
    float* gobj;                      // [cells] compact objectness gradient (NULL: forward only)
    float* row_tbox;                  // [cap*4] tbox of every row (aux output for the index-parity tests; may be NULL)
    unsigned long long inv_g2;        // ceil(2^40 / G2)
    float aw[ODPL_MAX_ANCHORS], ah[ODPL_MAX_ANCHORS];
    int* owner;                       // [cells]  -1 clean; max row index that maps here
    int* row_t; int* row_ag;          // [cap] target row ; (a << 4) | group
    int* row_cell;                    // [cap] linear cell or -1 (dropped)
    int* row_lead;                    // [cap] 1 if the row is the last row of its cell
    float* row_iou;                   // [cap]
    float* row_part;                  // [cap*2] (1 - iou), sum_k focal(cls)
    float* row_grad;                  // [cap*no]
    int* nrows;                       // [1]
    float* dense_part;                // [dense_nblk]
    int dense_blk0, dense_nblk;       // fill workgroups
    int conf_blk0, conf_nblk;         // objectness workgroups (thread = cell); dense_part has conf_nblk entries
    unsigned long long inv_no;        // ceil(2^40 / no)
};

struct VParams {
    Layer L[ODPL_MAX_SCALES];
    const float* target;
    int nl, B, na, nc, no, nT, cap, iou_mode, want_grad, bf16;
    const float* upstream;            // device scalar (backward) or NULL = 1
    int* status;
    float* out4; int* out_counts;
};

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kAT) void v5_assign_kernel(const VParams P) {
    const Layer& S = P.L[blockIdx.x];
    __shared__ int s_wcnt[kAT / 64];
    __shared__ int s_base;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, G = S.G, na = P.na, nT = P.nT;
    const float fG = (float)G;
    if (tid == 0) s_base = 0;
    __syncthreads();
    const int ncand = na * nT;
    for (int pass = 0; pass < 5; ++pass) {
        for (int c0 = 0; c0 < ncand; c0 += kAT) {
            const int i = c0 + tid;
            bool f = false; int a = 0, t = 0;
            if (i < ncand) {
                a = i / nT; t = i - a * nT;
                const float* y = P.target + (size_t)t * 6;
                const float gx = y[2] * fG, gy = y[3] * fG, gw = y[4] * fG, gh = y[5] * fG;     // accuracy.py:486
                const float rw = gw / S.aw[a], rh = gh / S.ah[a];                               // :488
                const float m = fmaxf(fmaxf(rw, 1.0f / rw), fmaxf(rh, 1.0f / rh));              // :489
                f = m < 4.0f;
                if (f && pass > 0) {                                                            // :503-504
                    const float v = (pass & 1) ? gx : gy;           // passes 1,3: x ; 2,4: y
                    const float fr = v - floorf(v);                 // torch.remainder(v, 1.)
                    f = pass <= 2 ? (fr < 0.5f && v > 1.0f) : (fr > 0.5f && v < fG - 1.0f);
                }
            }
            const unsigned long long mk = __ballot(f);
            if (lane == 0) s_wcnt[wv] = __popcll(mk);
            __syncthreads();
            int off = s_base, tot = 0;
            for (int w = 0; w < kAT / 64; ++w) { if (w < wv) off += s_wcnt[w]; tot += s_wcnt[w]; }
            if (f) {
                const int r = off + __popcll(mk & ((1ull << lane) - 1ull));
                S.row_t[r] = t; S.row_ag[r] = (a << 4) | pass;
            }
            __syncthreads();
            if (tid == 0) s_base += tot;
            __syncthreads();
        }
    }
    const int n = s_base;
    // cells + last-row-wins owner (losses.py:123 scatter; CPU index_put: sequential, last wins)
    int bad = 0;
    for (int r = tid; r < n; r += kAT) {
        const int t = S.row_t[r], ag = S.row_ag[r], a = ag >> 4, grp = ag & 15;
        const float* y = P.target + (size_t)t * 6;
        const float gx = y[2] * fG, gy = y[3] * fG;
        const float ox = grp == 1 ? 0.5f : (grp == 3 ? -0.5f : 0.0f), oy = grp == 2 ? 0.5f : (grp == 4 ? -0.5f : 0.0f);
        long long b = trunc_ll(y[0]), gi = trunc_ll(gx - ox), gj = trunc_ll(gy - oy);          // :509-513
        int cell = -1;
        if (wrap_idx(b, P.B) & wrap_idx(gj, G) & wrap_idx(gi, G)) cell = ((((int)b * na + a) * G) + (int)gj) * G + (int)gi;
        else bad = 1;                                               // the reference raises IndexError here
        S.row_cell[r] = cell;
        if (cell >= 0) atomicMax(&S.owner[cell], r);
    }
    bad = __syncthreads_or(bad);
    for (int r = tid; r < n; r += kAT) {
        const int cell = S.row_cell[r];
        S.row_lead[r] = (cell >= 0 && __hip_atomic_load(&S.owner[cell], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == r) ? 1 : 0;
    }
    for (int r = n + tid; r < P.cap; r += kAT) { S.row_cell[r] = -1; S.row_lead[r] = 0; }      // consumers of the compact form see "no row"
    if (tid == 0) { S.nrows[0] = n; if (bad) atomicOr(&P.status[0], 1); }
}

// ---------------------------------------------------------------------------------------------
// forward-mode value + 4 tangents (d/d px, py, pw, ph); tie / clamp conventions follow torch autograd
__device__ __forceinline__ float ld_head(const void* p, size_t i, int bf16) {
    return bf16 ? __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]) : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ void st_grad(void* p, size_t i, float v, int bf16) {
    if (bf16) reinterpret_cast<__hip_bfloat16*>(p)[i] = __float2bfloat16(v); else reinterpret_cast<float*>(p)[i] = v;
}
__device__ __forceinline__ unsigned short bf16_bits(float v) { const __hip_bfloat16 h = __float2bfloat16(v); return *reinterpret_cast<const unsigned short*>(&h); }

// element offset of (cell, channel c) in the head / gradient tensor
__device__ __forceinline__ size_t elem_off(const Layer& S, int no, int cell, int c) {
    if (S.layout == ODPL_LAYOUT_PLANAR) { const int plane = cell / S.G2, s = cell - plane * S.G2; return ((size_t)plane * no + c) * S.G2 + s; }
    return (size_t)cell * no + c;
}

struct D4 { float v; float d[4]; };
__device__ __forceinline__ D4 dconst(float c) { D4 r; r.v = c; r.d[0] = r.d[1] = r.d[2] = r.d[3] = 0.f; return r; }
__device__ __forceinline__ D4 dvar(float c, int i) { D4 r = dconst(c); r.d[i] = 1.f; return r; }
__device__ __forceinline__ D4 operator+(const D4& a, const D4& b) { D4 r; r.v = a.v + b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] + b.d[i]; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, const D4& b) { D4 r; r.v = a.v - b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] - b.d[i]; return r; }
__device__ __forceinline__ D4 operator*(const D4& a, const D4& b) { D4 r; r.v = a.v * b.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i]; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, const D4& b) { D4 r; r.v = a.v / b.v; for (int i = 0; i < 4; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) / b.v; return r; }
__device__ __forceinline__ D4 operator+(const D4& a, float c) { D4 r = a; r.v = a.v + c; return r; }
__device__ __forceinline__ D4 operator-(const D4& a, float c) { D4 r = a; r.v = a.v - c; return r; }
__device__ __forceinline__ D4 operator/(const D4& a, float c) { D4 r; r.v = a.v / c; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] / c; return r; }
__device__ __forceinline__ D4 rsub(float c, const D4& a) { D4 r; r.v = c - a.v; for (int i = 0; i < 4; ++i) r.d[i] = -a.d[i]; return r; }
__device__ __forceinline__ D4 dscale(const D4& a, float s) { D4 r; r.v = a.v; for (int i = 0; i < 4; ++i) r.d[i] = a.d[i] * s; return r; }
__device__ __forceinline__ D4 dmin(const D4& a, float c) { const float s = a.v < c ? 1.f : (a.v == c ? 0.5f : 0.f); D4 r = dscale(a, s); r.v = fminf(a.v, c); return r; }
__device__ __forceinline__ D4 dmax(const D4& a, float c) { const float s = a.v > c ? 1.f : (a.v == c ? 0.5f : 0.f); D4 r = dscale(a, s); r.v = fmaxf(a.v, c); return r; }
__device__ __forceinline__ D4 dclamp0(const D4& a) { D4 r = dscale(a, a.v >= 0.f ? 1.f : 0.f); r.v = fmaxf(a.v, 0.f); return r; }
__device__ __forceinline__ D4 dsq(const D4& a) { D4 r; r.v = a.v * a.v; for (int i = 0; i < 4; ++i) r.d[i] = 2.f * a.v * a.d[i]; return r; }

// accuracy.py:71-114 with x1y1x2y2=False; box1 = prediction (differentiated), box2 = target
__device__ D4 iou_family(float px, float py, float pw, float ph, float tx, float ty, float tw, float th, int mode) {
    const D4 X = dvar(px, 0), Y = dvar(py, 1), Wd = dvar(pw, 2), H = dvar(ph, 3);
    const D4 b1x1 = X - Wd / 2.0f, b1x2 = X + Wd / 2.0f, b1y1 = Y - H / 2.0f, b1y2 = Y + H / 2.0f;
    const float b2x1 = tx - tw / 2.0f, b2x2 = tx + tw / 2.0f, b2y1 = ty - th / 2.0f, b2y2 = ty + th / 2.0f;
    const D4 inter = dclamp0(dmin(b1x2, b2x2) - dmax(b1x1, b2x1)) * dclamp0(dmin(b1y2, b2y2) - dmax(b1y1, b2y1));
    const D4 w1 = b1x2 - b1x1, h1 = b1y2 - b1y1;
    const float w2 = b2x2 - b2x1, h2 = b2y2 - b2y1;
    const D4 uni = ((w1 * h1 + 1e-16f) + w2 * h2) - inter;
    const D4 iou = inter / uni;
    if (mode == 0) return iou;
    const D4 cw = dmax(b1x2, b2x2) - dmin(b1x1, b2x1), ch = dmax(b1y2, b2y2) - dmin(b1y1, b2y1);
    if (mode == 1) {
        const D4 c_area = cw * ch + 1e-16f;
        return iou - (c_area - uni) / c_area;
    }
    const D4 c2 = (dsq(cw) + dsq(ch)) + 1e-16f;
    const D4 rho2 = dsq(rsub(b2x1 + b2x2, b1x1 + b1x2)) / 4.0f + dsq(rsub(b2y1 + b2y2, b1y1 + b1y2)) / 4.0f;
    if (mode == 2) return iou - rho2 / c2;
    // CIoU: v = 4/pi^2 (atan(w2/h2) - atan(w1/h1))^2 ; alpha = v / (1 - iou + v) under no_grad
    const D4 q = w1 / h1;
    D4 at; at.v = atanf(q.v); { const float k = 1.0f / (1.0f + q.v * q.v); for (int i = 0; i < 4; ++i) at.d[i] = k * q.d[i]; }
    const D4 v = dscale(dsq(rsub(atanf(w2 / h2), at)), 1.0f);
    D4 vs; vs.v = 0.40528473456935109f * v.v; for (int i = 0; i < 4; ++i) vs.d[i] = 0.40528473456935109f * v.d[i];
    const float alpha = vs.v / (1.0f - iou.v + vs.v);
    D4 va; va.v = vs.v * alpha; for (int i = 0; i < 4; ++i) va.d[i] = vs.d[i] * alpha;
    return iou - (rho2 / c2 + va);
}

// FocalLoss(BCEWithLogitsLoss) element and its x-derivative (losses.py:47-57; gamma 1.5, alpha 0.25)
__device__ __forceinline__ void focal_bce(float x, float t, float& loss, float& dx) {
    const float bce = (1.0f - t) * x + (log1pf(expf(-fabsf(x))) + fmaxf(-x, 0.0f));
    const float p = sigmoidf_(x);
    const float pt = t * p + (1.0f - t) * (1.0f - p);
    const float af = t * 0.25f + (1.0f - t) * 0.75f;
    const float om = 1.0f - pt;
    const float mf = powf(om, 1.5f);
    loss = bce * (af * mf);
    const float dpt = (2.0f * t - 1.0f) * p * (1.0f - p);
    dx = af * ((p - t) * mf - bce * 1.5f * sqrtf(fmaxf(om, 0.0f)) * dpt);
}

__global__ __launch_bounds__(64) void v5_rows_kernel(const VParams P) {
    const Layer& S = P.L[blockIdx.y];
    const int r = blockIdx.x, lane = threadIdx.x, no = P.no, nc = P.nc;
    const int n = S.nrows[0];
    if (r >= n) return;
    const int cell = S.row_cell[r];
    float* rg = S.row_grad + (size_t)r * no;
    if (cell < 0) {
        for (int k = lane; k < no; k += 64) rg[k] = 0.f;
        if (lane == 0) { S.row_iou[r] = 0.f; S.row_part[r * 2] = 0.f; S.row_part[r * 2 + 1] = 0.f; }
        if (S.row_tbox && lane < 4) S.row_tbox[(size_t)r * 4 + lane] = 0.f;
        return;
    }
    const int t = S.row_t[r], ag = S.row_ag[r], a = ag >> 4, grp = ag & 15;
    auto ps = [&](int k) { return ld_head(S.p, elem_off(S, no, cell, k), P.bf16); };
    const float* y = P.target + (size_t)t * 6;
    const float fG = (float)S.G, fn = (float)n;
    // ---- box (lane 0): losses.py:115-119, accuracy.py:512-517
    if (lane == 0) {
        const float gx = y[2] * fG, gy = y[3] * fG, gw = y[4] * fG, gh = y[5] * fG;
        const float ox = grp == 1 ? 0.5f : (grp == 3 ? -0.5f : 0.0f), oy = grp == 2 ? 0.5f : (grp == 4 ? -0.5f : 0.0f);
        const float gi = (float)trunc_ll(gx - ox), gj = (float)trunc_ll(gy - oy);
        const float tx = gx - gi, ty = gy - gj;                     // tbox = (gxy - gij, gwh)
        if (S.row_tbox) { float* tb = S.row_tbox + (size_t)r * 4; tb[0] = tx; tb[1] = ty; tb[2] = gw; tb[3] = gh; }
        const float s0 = sigmoidf_(ps(0)), s1 = sigmoidf_(ps(1)), s2 = sigmoidf_(ps(2)), s3 = sigmoidf_(ps(3));
        const float px = s0 * 2.0f - 0.5f, py = s1 * 2.0f - 0.5f;
        const float u2 = s2 * 2.0f, u3 = s3 * 2.0f;
        const float pw = (u2 * u2) * S.aw[a], ph = (u3 * u3) * S.ah[a];
        const D4 io = iou_family(px, py, pw, ph, tx, ty, gw, gh, P.iou_mode);
        S.row_iou[r] = io.v;
        S.row_part[r * 2] = 1.0f - io.v;
        // d (0.05 * mean(1 - iou)) / d logits
        const float sc = -0.05f / fn;
        rg[0] = sc * io.d[0] * 2.0f * s0 * (1.0f - s0);
        rg[1] = sc * io.d[1] * 2.0f * s1 * (1.0f - s1);
        rg[2] = sc * io.d[2] * (2.0f * u2 * S.aw[a]) * 2.0f * s2 * (1.0f - s2);
        rg[3] = sc * io.d[3] * (2.0f * u3 * S.ah[a]) * 2.0f * s3 * (1.0f - s3);
        rg[4] = 0.f;
    }
    // ---- classes: losses.py:126-129
    float acc = 0.f;
    if (nc > 1) {
        const int tc = (int)trunc_ll(y[1]);
        const float sc = 0.58f / (fn * (float)nc);
        for (int k = lane; k < nc; k += 64) {
            float l, dx;
            focal_bce(ps(5 + k), k == tc ? 1.0f : 0.0f, l, dx);
            acc += l;
            rg[5 + k] = dx * sc;
        }
    } else {
        for (int k = lane; k < nc; k += 64) rg[5 + k] = 0.f;
    }
    acc = wave_sum(acc);
    if (lane == 0) S.row_part[r * 2 + 1] = acc;
}

// ---------------------------------------------------------------------------------------------
// obj pass: thread = cell.  objectness focal-BCE against tobj (the clamped IoU of the cell's owner row, 0 elsewhere),
// its gradient into the compact gobj, owner map reset.
__global__ __launch_bounds__(kDT) void v5_obj_kernel(const VParams P) {
    __shared__ float sh[kDT / 64];
    const int tid = threadIdx.x;
    int li = 0;
    for (int i = 0; i < P.nl; ++i) if ((int)blockIdx.x >= P.L[i].conf_blk0) li = i;
    const Layer& S = P.L[li];
    const int c = (blockIdx.x - S.conf_blk0) * kDT + tid;
    const int nrows_now = S.nrows[0];                         // rows the assign kernel of THIS call produced
    float acc = 0.f;
    if (c < S.cells) {
        int ow = S.owner[c];
        if (ow >= nrows_now) { atomicOr(P.status, 4); S.owner[c] = -1; ow = -1; }   // a dirty map (broken STATE_CLEAN promise) never indexes the row arrays
        const float tobj = ow >= 0 ? fmaxf(S.row_iou[ow], 0.0f) : 0.0f;             // losses.py:123 (gr = 1)
        float l, dx;
        focal_bce(ld_head(S.p, elem_off(S, P.no, c, 4), P.bf16), tobj, l, dx);
        acc = l;
        if (S.gobj) S.gobj[c] = dx * (1.0f / (float)S.cells);                       // mean over all cells, gain 1.0 (losses.py:135,138)
        if (ow >= 0) S.owner[c] = -1;                                               // leave the map clean
    }
    acc = wave_sum(acc);
    if ((tid & 63) == 0) sh[tid >> 6] = acc;
    __syncthreads();
    if (tid == 0) {
        float a = 0.f;
        for (int w = 0; w < kDT / 64; ++w) a += sh[w];
        S.dense_part[blockIdx.x - S.conf_blk0] = a;
    }
}

// ---------------------------------------------------------------------------------------------
// expand: the dense d loss/d p for a generic producer.  A workgroup owns a contiguous chunk of kChunkF4 16-byte quads of the
// gradient tensor; every quad is ONE non-temporal store (zeros, with gobj * upstream in the objectness slots).
template <bool BF16>
__global__ __launch_bounds__(kDT) void v5_expand_kernel(const VParams P) {
    constexpr int EPQ = BF16 ? 8 : 4;                        // elements per 16-byte quad
    const int tid = threadIdx.x;
    int li = 0;
    for (int i = 0; i < P.nl; ++i) if ((int)blockIdx.x >= P.L[i].dense_blk0) li = i;
    const Layer& S = P.L[li];
    const float f = P.upstream ? *P.upstream : 1.0f;
    const unsigned no = (unsigned)P.no, G2 = (unsigned)S.G2;
    const size_t total = (size_t)S.cells * no, totalq = total / EPQ;
    const size_t q0 = (size_t)(blockIdx.x - S.dense_blk0) * kChunkF4;
    const bool last = (q0 + kChunkF4) >= totalq;
    const size_t q1 = last ? totalq : (q0 + kChunkF4);
    // value of element e: gobj of its cell if it is an objectness slot, else 0
    auto elem = [&](size_t e) -> float {
        if (S.layout == ODPL_LAYOUT_PLANAR) {
            const unsigned row = (unsigned)(((unsigned long long)e * S.inv_g2) >> 40), s = (unsigned)e - row * G2;
            const unsigned plane = (unsigned)(((unsigned long long)row * S.inv_no) >> 40);
            return (row - plane * no == 4u) ? S.gobj[(size_t)plane * G2 + s] * f : 0.f;
        }
        const unsigned c = (unsigned)(((unsigned long long)e * S.inv_no) >> 40);
        return ((unsigned)e - c * no == 4u) ? S.gobj[c] * f : 0.f;
    };
    v4f* g4 = reinterpret_cast<v4f*>(S.grad);
    for (size_t q = q0 + tid; q < q1; q += kDT) {
        const size_t e = q * EPQ;
        float v[EPQ];
        bool any = false;
        if (S.layout == ODPL_LAYOUT_PLANAR) {
            // a quad inside one row: either all objectness (row % no == 4) or all zero
            const unsigned row = (unsigned)(((unsigned long long)e * S.inv_g2) >> 40), s = (unsigned)e - row * G2;
            if (s + EPQ <= G2) {
                const unsigned plane = (unsigned)(((unsigned long long)row * S.inv_no) >> 40);
                if (row - plane * no == 4u) {
                    any = true;
                    const float* src = S.gobj + (size_t)plane * G2 + s;
#pragma unroll
                    for (int k = 0; k < EPQ; ++k) v[k] = src[k] * f;
                }
            } else {
                any = true;
#pragma unroll
                for (int k = 0; k < EPQ; ++k) v[k] = elem(e + k);
            }
        } else {
            const unsigned c = (unsigned)(((unsigned long long)e * S.inv_no) >> 40);
            const unsigned r = (unsigned)e - c * no;
            // the quad can hold cell c's objectness slot (offset 4 - r) and, with 8 elements, cell c+1's (offset no + 4 - r)
            if (r <= 4u || r + EPQ > no + 4u) {
                any = true;
#pragma unroll
                for (int k = 0; k < EPQ; ++k) v[k] = 0.f;
                if (r <= 4u && 4u - r < (unsigned)EPQ) v[4 - r] = S.gobj[c] * f;
                const unsigned k2 = no + 4u - r;
                if (k2 < (unsigned)EPQ && (size_t)(c + 1) < (size_t)S.cells) v[k2] = S.gobj[c + 1] * f;
            }
        }
        v4f o = {0.f, 0.f, 0.f, 0.f};
        if (any) {
            if (!BF16) { o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3]; }
            else {
                unsigned* ob = reinterpret_cast<unsigned*>(&o);
#pragma unroll
                for (int k = 0; k < 4; ++k) ob[k] = bf16_bits(v[2 * k]) | ((unsigned)bf16_bits(v[2 * k + 1]) << 16);
            }
        }
        __builtin_nontemporal_store(o, g4 + q);
    }
    if (last && (size_t)tid < total - totalq * EPQ) {         // < EPQ tail elements of the layer
        const size_t e = totalq * EPQ + tid;
        st_grad(S.grad, e, elem(e), BF16);
    }
}

__global__ __launch_bounds__(64) void v5_cells_kernel(const VParams P) {
    // leader rows gather every row that landed in their cell (ascending row order) and write the sum over the zeros
    const Layer& S = P.L[blockIdx.y];
    const int r = blockIdx.x, lane = threadIdx.x, no = P.no;
    const int n = S.nrows[0];
    if (r >= n || !S.row_lead[r]) return;
    const int cell = S.row_cell[r];
    const float f = P.upstream ? *P.upstream : 1.0f;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;             // lane handles channels lane, lane+64, +128, +192
    for (int base = 0; base < n; base += 64) {
        const int rr = base + lane;
        const bool m = rr < n && S.row_cell[rr] == cell;
        unsigned long long mk = __ballot(m);
        while (mk) {
            const int j = __ffsll((long long)mk) - 1; mk &= mk - 1;
            const float* g = S.row_grad + (size_t)(base + j) * no;
            if (lane < no) a0 += g[lane];
            if (lane + 64 < no) a1 += g[lane + 64];
            if (lane + 128 < no) a2 += g[lane + 128];
            if (lane + 192 < no) a3 += g[lane + 192];
        }
    }
    if (lane < no && lane != 4) st_grad(S.grad, elem_off(S, no, cell, lane), a0 * f, P.bf16);
    if (lane + 64 < no) st_grad(S.grad, elem_off(S, no, cell, lane + 64), a1 * f, P.bf16);
    if (lane + 128 < no) st_grad(S.grad, elem_off(S, no, cell, lane + 128), a2 * f, P.bf16);
    if (lane + 192 < no) st_grad(S.grad, elem_off(S, no, cell, lane + 192), a3 * f, P.bf16);
}

__global__ __launch_bounds__(256) void v5_finalize_kernel(const VParams P) {
    __shared__ double sh[3 * 4];
    __shared__ float tot[3];
    const int tid = threadIdx.x;
    if (tid < 3) tot[tid] = 0.f;
    __syncthreads();
    for (int li = 0; li < P.nl; ++li) {
        const Layer& S = P.L[li];
        const int n = S.nrows[0];
        double a = 0, b = 0, c = 0;
        for (int r = tid; r < n; r += 256) { a += (double)S.row_part[r * 2]; b += (double)S.row_part[r * 2 + 1]; }
        for (int i = tid; i < S.conf_nblk; i += 256) c += (double)S.dense_part[i];
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_down(a, o, 64); b += __shfl_down(b, o, 64); c += __shfl_down(c, o, 64); }
        if ((tid & 63) == 0) { sh[(tid >> 6) * 3] = a; sh[(tid >> 6) * 3 + 1] = b; sh[(tid >> 6) * 3 + 2] = c; }
        __syncthreads();
        if (tid == 0) {
            const float sa = (float)(sh[0] + sh[3] + sh[6] + sh[9]), sb = (float)(sh[1] + sh[4] + sh[7] + sh[10]);
            const float sc = (float)(sh[2] + sh[5] + sh[8] + sh[11]);
            if (n > 0) {                                                     // losses.py:110: `if nb:`
                tot[0] += sa / (float)n;                                     // :119 mean(1 - iou)
                if (P.nc > 1) tot[1] += sb / ((float)n * (float)P.nc);       // :129
            }
            tot[2] += sc / (float)S.cells;                                   // :135
            if (P.out_counts) P.out_counts[li] = n;
        }
        __syncthreads();
    }
    if (tid == 0) {
        const float lbox = tot[0] * 0.05f, lobj = tot[2] * 1.0f, lcls = tot[1] * 0.58f;     // :137-139
        P.out4[0] = lbox + lobj + lcls; P.out4[1] = lbox; P.out4[2] = lcls; P.out4[3] = lobj;
        if (P.out_counts) P.out_counts[ODPL_MAX_SCALES] = P.status[0];
        P.status[0] = 0;
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct VPlan {
    // workspace: persistent owner maps + per-call scratch
    size_t status, owner[ODPL_MAX_SCALES], row_t[ODPL_MAX_SCALES], row_ag[ODPL_MAX_SCALES], row_iou[ODPL_MAX_SCALES],
        row_part[ODPL_MAX_SCALES], dense_part[ODPL_MAX_SCALES];
    // compact gradient (survives until backward; lives at the workspace's tail in a forward-only / one-call use)
    size_t gobj[ODPL_MAX_SCALES], row_grad[ODPL_MAX_SCALES], row_cell[ODPL_MAX_SCALES], row_lead[ODPL_MAX_SCALES],
        row_tbox[ODPL_MAX_SCALES], nrows[ODPL_MAX_SCALES], crow_t[ODPL_MAX_SCALES], crow_ag[ODPL_MAX_SCALES];
    size_t owner_bytes_total, ws_total, compact_total;
    int dense_nblk[ODPL_MAX_SCALES], conf_nblk[ODPL_MAX_SCALES], cap, no;
};

bool make_vplan(const odpl_yolov5_loss_desc_t* d, VPlan* p) {
    if (!d || d->nl < 1 || d->nl > ODPL_MAX_SCALES || d->B < 1 || d->na < 1 || d->na > ODPL_MAX_ANCHORS || d->nc < 1 || d->nT < 0) return false;
    if (d->iou_mode < 0 || d->iou_mode > 3 || (d->nT > 0 && !d->target)) return false;
    p->no = 5 + d->nc;
    if (p->no > 256) return false;
    p->cap = 5 * d->na * (d->nT > 0 ? d->nT : 1);
    size_t off = 0;
    p->status = off; off += 256;
    for (int i = 0; i < d->nl; ++i) {            // owner maps first, contiguous: one fill(0xFF) when not clean
        const odpl_yolov5_layer_t& l = d->layer[i];
        if (!l.p || l.G < 1) return false;
        if (l.layout != ODPL_LAYOUT_PLANAR && l.layout != ODPL_LAYOUT_INTERLEAVED) return false;
        const long long cells = (long long)d->B * d->na * l.G * l.G;
        if (cells * p->no > 0x7ffffff0LL) return false;
        p->owner[i] = off; off += align_up((size_t)cells * 4, 256);
    }
    p->owner_bytes_total = off - 256;
    size_t coff = 0;
    for (int i = 0; i < d->nl; ++i) {
        const size_t cap = (size_t)p->cap;
        const long long cells = (long long)d->B * d->na * d->layer[i].G * d->layer[i].G;
        p->dense_nblk[i] = (int)(((size_t)cells * p->no / (d->bf16 ? 8 : 4) + kChunkF4 - 1) / kChunkF4); if (p->dense_nblk[i] < 1) p->dense_nblk[i] = 1;
        p->conf_nblk[i] = (int)((cells + kDT - 1) / kDT);
        p->row_iou[i] = off;  off += align_up(cap * 4, 256);
        p->row_part[i] = off; off += align_up(cap * 8, 256);
        p->dense_part[i] = off; off += align_up((size_t)p->conf_nblk[i] * 4, 256);
        p->gobj[i] = coff;     coff += align_up((size_t)cells * 4, 256);
        p->row_grad[i] = coff; coff += align_up(cap * p->no * 4, 256);
        p->row_cell[i] = coff; coff += align_up(cap * 4, 256);
        p->row_lead[i] = coff; coff += align_up(cap * 4, 256);
        p->row_tbox[i] = coff; coff += align_up(cap * 16, 256);
        p->crow_t[i] = coff;   coff += align_up(cap * 4, 256);
        p->crow_ag[i] = coff;  coff += align_up(cap * 4, 256);
        p->nrows[i] = coff;    coff += 256;
    }
    p->ws_total = off;
    p->compact_total = coff;
    return true;
}

int fill_params(const odpl_yolov5_loss_desc_t* d, const VPlan& pl, char* ws, char* compact, VParams* out) {
    VParams& P = *out;
    P.target = d->target; P.nl = d->nl; P.B = d->B; P.na = d->na; P.nc = d->nc; P.no = pl.no; P.nT = d->nT; P.cap = pl.cap;
    P.bf16 = d->bf16 ? 1 : 0;
    P.iou_mode = d->iou_mode; P.status = ws ? (int*)(ws + pl.status) : nullptr; P.out4 = nullptr; P.out_counts = nullptr;
    P.upstream = nullptr; P.want_grad = 0;
    int dblk = 0, cblk = 0;
    for (int i = 0; i < d->nl; ++i) {
        const odpl_yolov5_layer_t& l = d->layer[i];
        Layer& S = P.L[i];
        if (((uintptr_t)l.p | (uintptr_t)l.grad) & 15) return ODPL_EINVAL;
        S.p = l.p; S.grad = l.grad; S.G = l.G; S.G2 = l.G * l.G; S.cells = d->B * d->na * l.G * l.G; S.layout = l.layout;
        for (int a = 0; a < ODPL_MAX_ANCHORS; ++a) { S.aw[a] = l.anchor_w[a]; S.ah[a] = l.anchor_h[a]; }
        if (ws) {
            S.owner = (int*)(ws + pl.owner[i]); S.row_iou = (float*)(ws + pl.row_iou[i]); S.row_part = (float*)(ws + pl.row_part[i]);
            S.dense_part = (float*)(ws + pl.dense_part[i]);
        } else { S.owner = nullptr; S.row_iou = nullptr; S.row_part = nullptr; S.dense_part = nullptr; }
        S.gobj = (float*)(compact + pl.gobj[i]); S.row_grad = (float*)(compact + pl.row_grad[i]);
        S.row_cell = (int*)(compact + pl.row_cell[i]); S.row_lead = (int*)(compact + pl.row_lead[i]);
        S.row_tbox = (float*)(compact + pl.row_tbox[i]); S.nrows = (int*)(compact + pl.nrows[i]);
        S.row_t = (int*)(compact + pl.crow_t[i]); S.row_ag = (int*)(compact + pl.crow_ag[i]);
        S.conf_blk0 = cblk; S.conf_nblk = pl.conf_nblk[i]; cblk += S.conf_nblk;
        S.dense_blk0 = dblk; S.dense_nblk = pl.dense_nblk[i]; dblk += S.dense_nblk;
        S.inv_no = ((1ull << 40) + (unsigned long long)pl.no - 1ull) / (unsigned long long)pl.no;
        S.inv_g2 = ((1ull << 40) + (unsigned long long)S.G2 - 1ull) / (unsigned long long)S.G2;
    }
    return 0;
}

int v5_forward(const odpl_yolov5_loss_desc_t* d, const VPlan& pl, char* ws, char* compact, float* out4, int32_t* out_counts, hipStream_t stream) {
    VParams P;
    const int rc = fill_params(d, pl, ws, compact, &P);
    if (rc) return rc;
    P.out4 = out4; P.out_counts = out_counts;
    int cblk = 0;
    for (int i = 0; i < d->nl; ++i) cblk += P.L[i].conf_nblk;
    if (!(d->flags & ODPL_YOLOV5_STATE_CLEAN)) {
        ODPL_HIP_TRY(odpl::fill_bytes(ws + pl.status, 0, 256, stream));
        ODPL_HIP_TRY(odpl::fill_bytes(ws + 256, 0xFF, pl.owner_bytes_total, stream));
    }
    hipLaunchKernelGGL(v5_assign_kernel, dim3(d->nl), dim3(kAT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    if (d->nT > 0) {
        hipLaunchKernelGGL(v5_rows_kernel, dim3(pl.cap, d->nl), dim3(64), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    prof_begin(ODPL_PROF_YOLOV5_OBJ, stream);
    hipLaunchKernelGGL(v5_obj_kernel, dim3(cblk), dim3(kDT), 0, stream, P);
    prof_end(ODPL_PROF_YOLOV5_OBJ, stream);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(v5_finalize_kernel, dim3(1), dim3(256), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

int v5_backward(const odpl_yolov5_loss_desc_t* d, const VPlan& pl, char* compact, const float* upstream, hipStream_t stream) {
    VParams P;
    const int rc = fill_params(d, pl, nullptr, compact, &P);
    if (rc) return rc;
    int want = 0, dblk = 0;
    for (int i = 0; i < d->nl; ++i) { if (d->layer[i].grad) want++; dblk += P.L[i].dense_nblk; }
    if (want == 0) return 0;
    if (want != d->nl) return ODPL_EINVAL;
    P.upstream = upstream; P.want_grad = 1;
    prof_begin(ODPL_PROF_YOLOV5_DENSE, stream);
    if (P.bf16) hipLaunchKernelGGL((v5_expand_kernel<true>), dim3(dblk), dim3(kDT), 0, stream, P);
    else hipLaunchKernelGGL((v5_expand_kernel<false>), dim3(dblk), dim3(kDT), 0, stream, P);
    prof_end(ODPL_PROF_YOLOV5_DENSE, stream);
    ODPL_HIP_TRY(hipGetLastError());
    if (d->nT > 0) {
        hipLaunchKernelGGL(v5_cells_kernel, dim3(pl.cap, d->nl), dim3(64), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}
}  // namespace

extern "C" size_t odpl_yolov5_loss_workspace_bytes(const odpl_yolov5_loss_desc_t* d) {
    VPlan p;
    return make_vplan(d, &p) ? p.ws_total + p.compact_total : 0;      // the one-call entry keeps the compact form at the workspace's tail
}

extern "C" size_t odpl_yolov5_loss_compact_bytes(const odpl_yolov5_loss_desc_t* d) {
    VPlan p;
    return make_vplan(d, &p) ? p.compact_total : 0;
}

extern "C" int odpl_yolov5_loss_forward(const odpl_yolov5_loss_desc_t* d, void* workspace, size_t workspace_bytes, void* compact,
                                        size_t compact_bytes, float* out4, int32_t* out_counts, void* stream) {
    VPlan pl;
    if (!make_vplan(d, &pl) || !workspace || !out4) return ODPL_EINVAL;
    char* ws = (char*)workspace;
    char* cp = (char*)compact;
    if (!cp) {                                                   // no gradient wanted: the compact scratch sits in the workspace
        if (workspace_bytes < pl.ws_total + pl.compact_total) return ODPL_EINVAL;
        cp = ws + pl.ws_total;
    } else if (workspace_bytes < pl.ws_total || compact_bytes < pl.compact_total) return ODPL_EINVAL;
    if (((uintptr_t)ws | (uintptr_t)cp) & 15) return ODPL_EINVAL;
    return v5_forward(d, pl, ws, cp, out4, out_counts, (hipStream_t)stream);
}

extern "C" int odpl_yolov5_loss_backward(const odpl_yolov5_loss_desc_t* d, const void* compact, size_t compact_bytes,
                                         const float* upstream, void* stream) {
    VPlan pl;
    if (!make_vplan(d, &pl) || !compact || compact_bytes < pl.compact_total || ((uintptr_t)compact & 15)) return ODPL_EINVAL;
    return v5_backward(d, pl, (char*)compact, upstream, (hipStream_t)stream);
}

extern "C" int odpl_yolov5_loss(const odpl_yolov5_loss_desc_t* d, void* workspace, size_t workspace_bytes,
                                float* out4, int32_t* out_counts, void* stream_) {
    VPlan pl;
    if (!make_vplan(d, &pl) || !workspace || workspace_bytes < pl.ws_total + pl.compact_total || !out4) return ODPL_EINVAL;
    if ((uintptr_t)workspace & 15) return ODPL_EINVAL;
    int want = 0;
    for (int i = 0; i < d->nl; ++i) if (d->layer[i].grad) want++;
    if (want != 0 && want != d->nl) return ODPL_EINVAL;
    char* ws = (char*)workspace;
    char* cp = ws + pl.ws_total;
    const int rc = v5_forward(d, pl, ws, cp, out4, out_counts, (hipStream_t)stream_);
    if (rc || !want) return rc;
    return v5_backward(d, pl, cp, nullptr, (hipStream_t)stream_);
}

extern "C" int odpl_yolov5_compact_view(const odpl_yolov5_loss_desc_t* d, const void* compact, int32_t layer, odpl_sparse_grad_t* out,
                                        odpl_yolov5_rows_t* rows) {
    VPlan pl;
    if (!make_vplan(d, &pl) || !compact || layer < 0 || layer >= d->nl) return ODPL_EINVAL;
    const char* c = (const char*)compact;
    if (out) {
        out->dense = (const float*)(c + pl.gobj[layer]);
        out->rows = (const float*)(c + pl.row_grad[layer]);
        out->row_cell = (const int32_t*)(c + pl.row_cell[layer]);
        out->n_rows = pl.cap;                                     // rows beyond the produced count carry row_cell = -1
        out->n_rows_dev = (const int32_t*)(c + pl.nrows[layer]);
        out->B = d->B; out->G2 = d->layer[layer].G * d->layer[layer].G; out->A = d->na; out->no = pl.no; out->dense_ch = 4;
        out->layout = ODPL_LAYOUT_PLANAR;                         // the CELL order (b*na + a)*G2 + s, whatever the memory format of p
    }
    if (rows) {
        rows->row_target = (const int32_t*)(c + pl.crow_t[layer]);
        rows->row_anchor_group = (const int32_t*)(c + pl.crow_ag[layer]);
        rows->row_cell = (const int32_t*)(c + pl.row_cell[layer]);
        rows->row_tbox = (const float*)(c + pl.row_tbox[layer]);
        rows->n_rows = (const int32_t*)(c + pl.nrows[layer]);
        rows->cap = pl.cap;
    }
    return 0;
}
