// Evaluation statistics on the device: get_batch_statistics (accuracy.py:116-154) and
// ap_per_class / compute_ap (accuracy.py:207-287).  Integer / index work is exact; the curves are
// double precision like the reference's numpy arrays.
#include "odpl_common.hpp"
#include "odpl_boxes_dev.hpp"
#include "odpl_sort_dev.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kAnnCap = 1024;

// ---- get_batch_statistics: one wave per image, predictions visited in their given order
__global__ __launch_bounds__(64) void batch_stats_kernel(const float* __restrict__ dets, const int* __restrict__ counts,
                                                         const float* __restrict__ targets, int nT, int Kmax, float thr,
                                                         float* __restrict__ tp, int* __restrict__ status) {
    __shared__ float4 s_box[kAnnCap];
    __shared__ float s_lab[kAnnCap];
    __shared__ unsigned char s_used[kAnnCap];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int K = counts[b];
    float* tpo = tp + (size_t)b * Kmax;
    for (int k = lane; k < Kmax; k += 64) tpo[k] = 0.f;
    if (K <= 0) return;
    // annotations of this image in row order (accuracy.py:132)
    int na = 0;
    for (int base = 0; base < nT; base += 64) {
        const int t = base + lane;
        const bool m = t < nT && targets[(size_t)t * 6] == (float)b;
        const unsigned long long mk = __ballot(m);
        const int pos = na + __popcll(mk & ((1ull << lane) - 1ull));
        if (m && pos < kAnnCap) {
            const float* y = targets + (size_t)t * 6;
            s_box[pos] = make_float4(y[2], y[3], y[4], y[5]); s_lab[pos] = y[1]; s_used[pos] = 0;
        }
        na += __popcll(mk);
    }
    if (na > kAnnCap) { if (lane == 0) atomicOr(status, 1); na = kAnnCap; }
    __syncthreads();
    if (na == 0) return;
    int nd = 0;
    const float* d = dets + (size_t)b * Kmax * 7;
    for (int k = 0; k < K; ++k) {
        if (nd == na) break;                                        // accuracy.py:141
        const float lab = d[k * 7 + 6];
        Box4 pb; pb.x1 = d[k * 7]; pb.y1 = d[k * 7 + 1]; pb.x2 = d[k * 7 + 2]; pb.y2 = d[k * 7 + 3];
        bool any = false; float bv = 0.f; int bi = 0x7fffffff; bool have = false;
        for (int j = lane; j < na; j += 64) {
            any |= (s_lab[j] == lab);
            const float4 q = s_box[j];
            Box4 tb; tb.x1 = q.x; tb.y1 = q.y; tb.x2 = q.z; tb.y2 = q.w;
            const float v = iou_plus1(pb, tb);                      // accuracy.py:148
            if (!have || (bv == bv && (v > bv || v != v))) { bv = v; bi = j; have = true; }
        }
        if (!__ballot(any)) continue;                               // accuracy.py:145
        if (!have) { bv = -INFINITY; bi = 0x7fffffff; }
        for (int o = 32; o > 0; o >>= 1) {
            const float ov = __shfl_down(bv, o, 64); const int oi = __shfl_down(bi, o, 64);
            const bool onan = ov != ov, bnan = bv != bv;
            const bool take = (onan != bnan) ? onan : ((!onan && ov != bv) ? ov > bv : oi < bi);
            if (take) { bv = ov; bi = oi; }
        }
        bv = __shfl(bv, 0, 64); bi = __shfl(bi, 0, 64);
        if (bv >= thr && !s_used[bi]) {                             // accuracy.py:149-151
            if (lane == 0) { tpo[k] = 1.f; s_used[bi] = 1; }
            nd++;
        }
        __syncthreads();
    }
}

// ---- ap_per_class
struct ApParams {
    const double* tp; const float* conf; const float* pred_cls; const float* target_cls;
    int N, Npad, M, Mpad;
    float* key; unsigned* idx;                 // [Npad] predictions sorted by conf desc
    float* tkey; unsigned* tidx;               // [Mpad] targets sorted by class
    float* classes; int* n_gt; int* n_cls;     // [M], [M], [1]
    int* region_off;                           // [M+1]
    unsigned char* r_tp;                       // [N] tp flag of the class-compacted, conf-sorted predictions
    double* r_prec; double* r_rec;             // [N]
    double* out_p; double* out_r; double* out_ap; double* out_f1; int* out_cls;
};

__global__ __launch_bounds__(1024) void ap_prepare_kernel(const ApParams P) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // predictions: argsort(-conf) (accuracy.py:220); key = -conf ascending == (conf desc, idx asc) here
    for (int i = tid; i < P.Npad; i += 1024) { P.key[i] = i < P.N ? P.conf[i] : -INFINITY; P.idx[i] = i < P.N ? (unsigned)i : 0xffffffffu; }
    for (int i = tid; i < P.Mpad; i += 1024) { P.tkey[i] = i < P.M ? -P.target_cls[i] : -INFINITY; P.tidx[i] = i < P.M ? (unsigned)i : 0xffffffffu; }
    __syncthreads();
    bitonic_sort_global(P.key, P.idx, P.Npad);
    bitonic_sort_global(P.tkey, P.tidx, P.Mpad);             // -class desc == class asc (np.unique order)
    // unique classes + counts
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < P.M; base += 1024) {
        const int i = base + tid;
        const bool first = i < P.M && (i == 0 || P.tkey[i] != P.tkey[i - 1]);
        const unsigned long long mk = __ballot(first);
        if (lane == 0) s_w[wv] = __popcll(mk);
        __syncthreads();
        int off = s_carry, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) off += s_w[w]; tot += s_w[w]; }
        if (first) {
            const int c = off + __popcll(mk & ((1ull << lane) - 1ull));
            P.classes[c] = -P.tkey[i];
            int e = i + 1; while (e < P.M && P.tkey[e] == P.tkey[i]) ++e;      // run length = n_gt
            P.n_gt[c] = e - i;
        }
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    const int nc = s_carry;
    if (tid == 0) P.n_cls[0] = nc;
    // region offsets: predictions per class, exclusive scan in class order (single thread: nc is small)
    __syncthreads();
    for (int c = tid; c < nc; c += 1024) {
        int n = 0; const float cv = P.classes[c];
        for (int i = 0; i < P.N; ++i) n += (P.pred_cls[i] == cv);
        P.region_off[c + 1] = n;
    }
    __syncthreads();
    if (tid == 0) { P.region_off[0] = 0; for (int c = 0; c < nc; ++c) P.region_off[c + 1] += P.region_off[c]; }
}

__global__ __launch_bounds__(1024) void ap_class_kernel(const ApParams P) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    __shared__ double s_d[16];
    __shared__ double s_dc;
    const int c = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (c >= P.n_cls[0]) return;
    const float cv = P.classes[c];
    const int r0 = P.region_off[c], np_ = P.region_off[c + 1] - r0, ngt = P.n_gt[c];
    if (tid == 0) { P.out_cls[c] = (int)cv; }
    if (np_ == 0) {                                           // accuracy.py:236-239 (n_gt >= 1 always here)
        if (tid == 0) { P.out_p[c] = 0; P.out_r[c] = 0; P.out_ap[c] = 0; P.out_f1[c] = 0; }
        return;
    }
    // (1) stable compaction of this class's predictions in conf order
    if (tid == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < P.N; base += 1024) {
        const int i = base + tid;
        unsigned src = 0; bool m = false;
        if (i < P.N) { src = P.idx[i]; m = P.pred_cls[src] == cv; }
        const unsigned long long mk = __ballot(m);
        if (lane == 0) s_w[wv] = __popcll(mk);
        __syncthreads();
        int off = s_carry, tot = 0;
        for (int w = 0; w < 16; ++w) { if (w < wv) off += s_w[w]; tot += s_w[w]; }
        if (m) P.r_tp[r0 + off + __popcll(mk & ((1ull << lane) - 1ull))] = P.tp[src] != 0.0 ? 1 : 0;
        __syncthreads();
        if (tid == 0) s_carry += tot;
        __syncthreads();
    }
    // (2) cumulative TP count -> precision / recall curves (accuracy.py:242-250); counts are exact integers
    if (tid == 0) s_carry = 0;
    __syncthreads();
    const double dgt = (double)ngt + 1e-16;
    for (int base = 0; base < np_; base += 1024) {
        const int i = base + tid;
        const int t = i < np_ ? P.r_tp[r0 + i] : 0;
        int v = t;
        for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(v, o, 64); if (lane >= o) v += u; }
        if (lane == 63) s_w[wv] = v;
        __syncthreads();
        int woff = s_carry;
        for (int w = 0; w < wv; ++w) woff += s_w[w];
        const int tpc = woff + v;
        if (i < np_) { P.r_prec[r0 + i] = (double)tpc / (double)(i + 1); P.r_rec[r0 + i] = (double)tpc / dgt; }
        __syncthreads();
        if (tid == 1023) s_carry = tpc;
        __syncthreads();
    }
    const double p_last = P.r_prec[r0 + np_ - 1], r_last = P.r_rec[r0 + np_ - 1];
    // (3) precision envelope (suffix max, sentinel 0 behind the last point) and the area under it (accuracy.py:274-286)
    if (tid == 0) s_dc = 0.0;
    __syncthreads();
    double area = 0.0;
    for (int hi = np_; hi > 0; hi -= 1024) {
        const int i = hi - 1 - tid;                           // reverse order: thread 0 takes the last element
        double v = i >= 0 ? P.r_prec[r0 + i] : 0.0;
        for (int o = 1; o < 64; o <<= 1) { const double u = __shfl_up(v, o, 64); if (lane >= o) v = fmax(v, u); }
        if (lane == 63) s_d[wv] = v;
        __syncthreads();
        double carry = s_dc;
        for (int w = 0; w < wv; ++w) carry = fmax(carry, s_d[w]);
        const double env = fmax(v, carry);
        if (i >= 0) {
            const double r = P.r_rec[r0 + i], rp = i > 0 ? P.r_rec[r0 + i - 1] : 0.0;
            if (r != rp) area += (r - rp) * env;
        }
        __syncthreads();
        if (tid == 1023) s_dc = env;
        __syncthreads();
    }
    for (int o = 32; o > 0; o >>= 1) area += __shfl_down(area, o, 64);
    if (lane == 0) s_d[wv] = area;
    __syncthreads();
    if (tid == 0) {
        double a = 0.0;
        for (int w = 0; w < 16; ++w) a += s_d[w];
        P.out_p[c] = p_last; P.out_r[c] = r_last; P.out_ap[c] = a;
        P.out_f1[c] = 2.0 * p_last * r_last / (p_last + r_last + 1e-16);      // accuracy.py:258
    }
}

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }
inline int next_pow2(int v) { int p = 2; while (p < v) p <<= 1; return p; }
}  // namespace

extern "C" int odpl_batch_statistics(const float* dets, const int32_t* counts, const float* targets, int32_t B, int32_t Kmax,
                                     int32_t nT, float iou_threshold, float* tp, int32_t* status, void* stream) {
    if (!dets || !counts || !tp || !status || B < 1 || Kmax < 1 || nT < 0 || (nT > 0 && !targets)) return ODPL_EINVAL;
    ODPL_HIP_TRY(odpl::fill_bytes(status, 0, sizeof(int), (hipStream_t)stream));
    hipLaunchKernelGGL(batch_stats_kernel, dim3(B), dim3(64), 0, (hipStream_t)stream, dets, counts, targets, nT, Kmax, iou_threshold, tp, status);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" size_t odpl_ap_per_class_workspace_bytes(int32_t N, int32_t M) {
    if (N < 1 || M < 1 || N > (1 << 24) || M > (1 << 24)) return 0;
    const size_t np = next_pow2(N), mp = next_pow2(M);
    return align_up(np * 8, 256) + align_up(mp * 8, 256) + align_up((size_t)M * 8, 256) + 256 + align_up(((size_t)M + 1) * 4, 256) +
           align_up((size_t)N, 256) + 2 * align_up((size_t)N * 8, 256) + 1024;
}

extern "C" int odpl_ap_per_class(const double* tp, const float* conf, const float* pred_cls, const float* target_cls,
                                 int32_t N, int32_t M, void* workspace, size_t workspace_bytes,
                                 double* out_p, double* out_r, double* out_ap, double* out_f1, int32_t* out_classes,
                                 int32_t* out_num_classes, void* stream_) {
    const size_t need = odpl_ap_per_class_workspace_bytes(N, M);
    if (!need || !tp || !conf || !pred_cls || !target_cls || !workspace || workspace_bytes < need || !out_p || !out_r || !out_ap ||
        !out_f1 || !out_classes || !out_num_classes || ((uintptr_t)workspace & 15))
        return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    ApParams P;
    P.tp = tp; P.conf = conf; P.pred_cls = pred_cls; P.target_cls = target_cls; P.N = N; P.M = M;
    P.Npad = next_pow2(N); P.Mpad = next_pow2(M);
    char* w = (char*)workspace; size_t off = 0;
    P.key = (float*)(w + off); off += align_up((size_t)P.Npad * 4, 256);
    P.idx = (unsigned*)(w + off); off += align_up((size_t)P.Npad * 4, 256);
    P.tkey = (float*)(w + off); off += align_up((size_t)P.Mpad * 4, 256);
    P.tidx = (unsigned*)(w + off); off += align_up((size_t)P.Mpad * 4, 256);
    P.classes = (float*)(w + off); off += align_up((size_t)M * 4, 256);
    P.n_gt = (int*)(w + off); off += align_up((size_t)M * 4, 256);
    P.region_off = (int*)(w + off); off += align_up(((size_t)M + 1) * 4, 256);
    P.r_tp = (unsigned char*)(w + off); off += align_up((size_t)N, 256);
    P.r_prec = (double*)(w + off); off += align_up((size_t)N * 8, 256);
    P.r_rec = (double*)(w + off); off += align_up((size_t)N * 8, 256);
    P.n_cls = out_num_classes;
    P.out_p = out_p; P.out_r = out_r; P.out_ap = out_ap; P.out_f1 = out_f1; P.out_cls = out_classes;
    hipLaunchKernelGGL(ap_prepare_kernel, dim3(1), dim3(1024), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(ap_class_kernel, dim3(M), dim3(1024), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
