// BatchNorm2d(train) + activation in ONE launch per direction, for the maps the chip can hold on its own registers.
//
// The three-launch form (odpl_bnact.hip) reads the convolution's output twice in forward (statistics, then apply) and reads
// (grad_y, x) twice in backward (parameter gradients, then dx): the statistics are a grid-wide dependency between the two
// passes.  On the small layers of the BASELINE networks (YOLOv4: 90 of 107 BatchNorm layers are <= 44 MB) the passes themselves are
// launch / ramp bound: six dependent launches of 5-15 us per layer.  Here a workgroup keeps its tile -- KR rows x one channel slice
// per lane, 32-64 KB per workgroup, <= 45 MB per launch -- in REGISTERS across the dependency:
//   phase 1   load the tile once; per-channel partial sums in double; publish them ([tile][C][2], write-through stores)
//   exchange  the workgroups of one channel slice (T of them, a few dozen) meet on one counter: arrive, poll, one acquire.  NOT a grid
//             barrier: slices are independent, and a workgroup needs only the 16-45 KB of partials of its own slice
//   phase 2   every workgroup reduces its slice's partials in the same fixed order (bitwise the same statistics everywhere; tile 0
//             writes them out), then applies them to the tile it still holds: forward 1 read + 1 write instead of 2 + 1, backward
//             2 reads + 1 write instead of 4 + 1, activation derivative evaluated once instead of twice, one launch instead of three.
// Inter-workgroup visibility follows the agent-scope rules of gfx950 (per-XCD L2s are not coherent with each other, a CU's L1 is never
// refreshed by other CUs' stores): partials are stored write-through (sc1: relaxed agent-scope atomic stores of 8 bytes) and drained
// by every storing wave (s_waitcnt vmcnt(0)) before the workgroup barrier that precedes ONE lane's arrival add; the consumer polls
// that counter relaxed from one lane, then ONE agent-scope acquire + s_waitcnt + workgroup barrier, and reads the partials with
// sc1 loads.  Nothing depends on dispatch order or workgroup placement.
// Progress does not depend on co-residency either: the wait is BOUNDED (100 MHz clock).  A workgroup whose peers have not arrived
// in time (another process sharing the GPU, a grid larger than the free slots) computes the whole slice's statistics itself from
// global memory -- the same arithmetic in the same order, hence the same bits -- and goes on; nobody waits for it.  The host keeps
// the grid within the occupancy the runtime reports (minus a margin) so that this path is never the normal one.
#include "odpl_bn_dev.hpp"

#include <atomic>

// MEASURED (MI355X, round 3, profiles/r03_bnfused_*.txt): correct -- bitwise reproducible, the self-service path gives the same bits --
// but NOT faster, so it is OFF by default (odpl_tune_set(ODPL_TUNE_BN_FUSED, 1) turns it on).  The in-launch exchange costs 15-25 us with
// 688 workgroups in flight: write-through partials take 3-16 us to drain while the other workgroups still stream their tiles, everybody
// then waits for the slowest drain, the acquire costs 2-3.5 us at 2.7 workgroups per CU and the write-through partials come back from
// memory in ~6 us -- against 4.8 us for the finalize launch + two launch boundaries + one re-read of a map that is still in the
// 256 MB Infinity Cache.  Per layer, device time, forward / backward: 5.5 MB 16.8 / 16.6 us (three launches: 16.2 / 18.4),
// 11 MB 22-24 / 24-26 (18.3 / 23.3), 22 MB 45 / 45 (22.7 / 31.1), 44 MB forward 63 (32.6).
namespace odpl {
int g_bn_fused = 0;
int g_bn_fused_timeout_us = 20000;
}

namespace {
using namespace odpl;
constexpr int kT = 256;
constexpr int kSlots = 64;              // concurrent calls (streams, graph replays in flight) each own a counter row
constexpr int kMaxSlices = 64;
constexpr int kMaxChunk = 24;           // tiles one thread of the phase-2 reduction sums (all of their partials in flight at once)

typedef __attribute__((address_space(1))) unsigned long long gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ unsigned g_bn_sync[kSlots * kMaxSlices * 2];    // zero at load; the last workgroup to leave a slice group resets its pair
__device__ unsigned g_bn_fallbacks;                        // diagnostic: workgroups that took the self-service path
constexpr int kStampWgs = 2048;
__device__ unsigned long long g_bn_stamp[kStampWgs * 8];   // diagnostic: phase boundaries of the last stamped launch
bool g_stamp_on = false;

__device__ __forceinline__ void st_u64_sc1(double* p, double v) {
    __hip_atomic_store((gu64*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_f64_sc1(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((gu64*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}

// store 16 / 8 bytes of a channels-last row at a 32-bit byte offset from a workgroup-uniform base (scalar base + one offset register)
template <bool BF>
__device__ __forceinline__ void st_quad_at(char* base, unsigned byte_off, const float4& v) {
    if (!BF) { *reinterpret_cast<float4*>(base + byte_off) = v; return; }
    uint2 u;
    u.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16); u.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
    *reinterpret_cast<uint2*>(base + byte_off) = u;
}

// Per-channel sums of one row tile for the channel quad q of this thread: rows tr + u * rows_par, u ascending, then the row lanes
// combined through LDS in lane order.  The owner of a tile (KEEP: the tile stays in hx / hg) and a workgroup that recomputes a peer's
// tile (self-service: one row at a time, nothing kept) run the SAME additions in the SAME order.
//   forward : a = sum x, b = sum x^2;            hx = x
//   backward: a = sum gz, b = sum gz * xhat;     hx = xhat, hg = gz        (gz = gy * act'(gamma * xhat + beta))
template <bool BWD, bool BF, int ACT>
struct TileCtx {
    const char* bx; const char* bg; unsigned xs, gs; int nrows;
    float4 mu, is, ga, be;
};

template <bool BWD, bool BF, int ACT>
__device__ __forceinline__ void tile_elem(const TileCtx<BWD, BF, ACT>& c, float4& v, float4& g, double (&a)[4], double (&b)[4]) {
    if (!BWD) {
        a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
        b[0] = fma((double)v.x, (double)v.x, b[0]); b[1] = fma((double)v.y, (double)v.y, b[1]);
        b[2] = fma((double)v.z, (double)v.z, b[2]); b[3] = fma((double)v.w, (double)v.w, b[3]);
    } else {
        const float xh0 = (v.x - c.mu.x) * c.is.x, xh1 = (v.y - c.mu.y) * c.is.y, xh2 = (v.z - c.mu.z) * c.is.z, xh3 = (v.w - c.mu.w) * c.is.w;
        const float g0 = g.x * act_grad<ACT>(c.ga.x * xh0 + c.be.x), g1 = g.y * act_grad<ACT>(c.ga.y * xh1 + c.be.y);
        const float g2 = g.z * act_grad<ACT>(c.ga.z * xh2 + c.be.z), g3 = g.w * act_grad<ACT>(c.ga.w * xh3 + c.be.w);
        a[0] += g0; a[1] += g1; a[2] += g2; a[3] += g3;
        b[0] = fma((double)g0, (double)xh0, b[0]); b[1] = fma((double)g1, (double)xh1, b[1]);
        b[2] = fma((double)g2, (double)xh2, b[2]); b[3] = fma((double)g3, (double)xh3, b[3]);
        v = make_float4(xh0, xh1, xh2, xh3); g = make_float4(g0, g1, g2, g3);
    }
}

// combine the row lanes (fixed order); valid in the threads with tr == 0
__device__ __forceinline__ void combine_rows(double* s_red, int tr, int tc, int tpr, int rows_par, double (&a)[4], double (&b)[4]) {
    __syncthreads();
    double* d = s_red + ((size_t)tr * tpr + tc) * 8;
#pragma unroll
    for (int i = 0; i < 4; ++i) { d[i] = a[i]; d[4 + i] = b[i]; }
    __syncthreads();
    if (tr == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) { a[i] = 0; b[i] = 0; }
        for (int t = 0; t < rows_par; ++t) {
            const double* e = s_red + ((size_t)t * tpr + tc) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += e[i]; b[i] += e[4 + i]; }
        }
    }
}

// Waves per SIMD the register allocation must leave room for (= workgroups of 256 threads per CU): the tile is 4 KR (forward) or
// 8 KR (backward: xhat and gz) registers per lane; what counts is that the whole grid is resident at once.
constexpr int min_waves(bool bwd, int kr) { return (bwd ? 2 : 1) * kr <= 8 ? 4 : 3; }

template <bool BWD, bool BF, int ACT, int KR>
__global__ __launch_bounds__(kT, min_waves(BWD, KR)) void bn_fused_kernel(const BnParams P) {
    constexpr unsigned QB = BF ? 8u : 16u;
    extern __shared__ __align__(16) double s_mem[];
    const int C = P.C, c4n = C >> 2;
    const int s = P.slice_q, tpr = s, rows_par = kT / s;              // s quads (4 s channels) per workgroup; s in {8, 16}
    const int tr = threadIdx.x / tpr, tc = threadIdx.x - tr * tpr;
    const int tile = blockIdx.x, slice = blockIdx.y, T = P.nblk;
    const int q = slice * s + tc;                                     // < c4n: c4n % s == 0
    double* s_red = s_mem;                                            // [kT][8]
    double* s_tot = s_mem + kT * 8;                                   // [4 s][2]  the slice's sums over all tiles
    float* s_stat = reinterpret_cast<float*>(s_tot + 384);           // [4 s][2]  (mean, invstd) / (dbeta, dgamma); s_tot + 256: self-service scratch
    int* s_flag = reinterpret_cast<int*>(s_stat + 128);

    TileCtx<BWD, BF, ACT> c;
    if (BWD) {
        c.mu = reinterpret_cast<const float4*>(P.mean)[q]; c.is = reinterpret_cast<const float4*>(P.invstd)[q];
        c.ga = reinterpret_cast<const float4*>(P.gamma)[q]; c.be = reinterpret_cast<const float4*>(P.beta)[q];
    }
    auto set_tile = [&](int t) {
        const long long r0 = (long long)t * P.rows_per_blk;
        long long r1 = r0 + P.rows_per_blk; if (r1 > P.R) r1 = P.R;
        c.nrows = (int)(r1 - r0);
        c.bx = reinterpret_cast<const char*>(P.x) + (size_t)r0 * c4n * QB;
        c.bg = BWD ? reinterpret_cast<const char*>(P.gy) + (size_t)r0 * (size_t)P.gy_rs4 * QB : nullptr;
        c.xs = (unsigned)c4n * QB; c.gs = (unsigned)P.gy_rs4 * QB;
    };

    const int wg = blockIdx.y * gridDim.x + blockIdx.x;
    auto stamp = [&](int i) { if (P.stamp && threadIdx.x == 0 && wg < kStampWgs) P.stamp[wg * 8 + i] = __builtin_amdgcn_s_memrealtime(); };
    stamp(0);
    // ---- phase 1: this workgroup's tile, loaded once and kept
    set_tile(tile);
    const int my_rows = c.nrows;
    float4 hx[KR], hg[BWD ? KR : 1];
#pragma unroll
    for (int u = 0; u < KR; ++u) {                                    // unconditional, clamped: all KR (x 2) loads in flight at once
        int r = tr + u * rows_par; if (r > my_rows - 1) r = my_rows - 1;
        hx[u] = ld_quad_at<BF>(c.bx, (unsigned)r * c.xs + (unsigned)q * QB);
        if (BWD) hg[u] = ld_quad_at<BF>(c.bg, (unsigned)r * c.gs + (unsigned)q * QB);
    }
    double a[4] = {0, 0, 0, 0}, b[4] = {0, 0, 0, 0};
#pragma unroll
    for (int u = 0; u < KR; ++u) {
        if (tr + u * rows_par < my_rows) tile_elem<BWD, BF, ACT>(c, hx[u], hg[BWD ? u : 0], a, b);
        __builtin_amdgcn_sched_barrier(0);                            // one row at a time: the scheduler otherwise widens every row to double first
    }
    combine_rows(s_red, tr, tc, tpr, rows_par, a, b);
    stamp(1);
    if (tr == 0) {
        double* o = P.part + ((size_t)tile * C + 4 * q) * 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) { st_u64_sc1(o + 2 * i, a[i]); st_u64_sc1(o + 2 * i + 1, b[i]); }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // every storing wave drains its write-through stores ...
    __syncthreads();                                                  // ... before the one lane that signals for all of them
    gu32* arrive = (gu32*)(P.sync + 2 * slice);
    gu32* depart = arrive + 1;
    stamp(2);
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // ---- exchange: ONE lane polls ONE word, relaxed, with a bound
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int ok = 1;
        while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)T) {
            __builtin_amdgcn_s_sleep(24);                             // ~0.6 us: hundreds of lanes poll a handful of words -- keep that traffic off the channels the late tiles load from
            if (__builtin_amdgcn_s_memrealtime() - t0 > (unsigned long long)P.timeout_ticks) { ok = 0; break; }
        }
        *s_flag = ok;
        stamp(3);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");            // ONE acquire after the match
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();
    stamp(4);
    const bool all_here = *s_flag != 0;

    // ---- the slice's sums over its T tiles: each (channel, which) value is summed by G = 256 / (8 s) threads, thread g taking the
    //      g-th contiguous chunk of tiles in tile order; the G chunk sums are then added in chunk order
    const int nval = 8 * s, G = kT / nval, chunk = (T + G - 1) / G;
    double* s_chunk = s_red;                                          // [G][nval] (s_red is free again)
    if (all_here) {
        const int j = threadIdx.x % nval, g = threadIdx.x / nval;
        const int ch = slice * 4 * s + (j >> 1), which = j & 1;
        const int k0 = g * chunk, k1 = (g + 1) * chunk < T ? (g + 1) * chunk : T;
        const double* base = P.part + (size_t)ch * 2 + which;
        // every load of the chunk is issued before the first addition (one trip to memory, not chunk / 4 of them): unconditional,
        // clamped to the chunk's last tile; the additions then run in tile order, the clamped repeats counting as +0
        double v[kMaxChunk];
#pragma unroll
        for (int u = 0; u < kMaxChunk; ++u) {
            int k = k0 + u; if (k > k1 - 1) k = k1 - 1; if (k < 0) k = 0;
            v[u] = ld_f64_sc1(base + (size_t)k * C * 2);
        }
        double t = 0;
#pragma unroll
        for (int u = 0; u < kMaxChunk; ++u) t += (k0 + u < k1) ? v[u] : 0.0;
        s_chunk[g * nval + j] = t;
        __syncthreads();
        if (g == 0) {
            double tot = s_chunk[j];
            for (int gg = 1; gg < G; ++gg) tot += s_chunk[gg * nval + j];
            s_tot[j] = tot;
        }
    } else {
        // self-service: recompute every tile of the slice from global memory, one row per lane at a time -- the same additions in the
        // same order as the owners' (per tile), the chunk readers' (per chunk) and the final combine.  Running sums live in LDS
        // (s_tot: totals, s_acc: the current chunk): this path must not cost the normal one registers.
        if (threadIdx.x == 0) atomicAdd(&g_bn_fallbacks, 1u);
        double* s_acc = s_tot + 256;                                  // [4 s][2]
        if (threadIdx.x < nval) { s_tot[threadIdx.x] = 0; s_acc[threadIdx.x] = 0; }
        for (int k = 0; k < T; ++k) {
            set_tile(k);
            double pa[4] = {0, 0, 0, 0}, pb[4] = {0, 0, 0, 0};
            for (int u = 0; u < KR; ++u) {
                const int r = tr + u * rows_par;
                if (r < c.nrows) {
                    float4 v = ld_quad_at<BF>(c.bx, (unsigned)r * c.xs + (unsigned)q * QB), g = v;
                    if (BWD) g = ld_quad_at<BF>(c.bg, (unsigned)r * c.gs + (unsigned)q * QB);
                    tile_elem<BWD, BF, ACT>(c, v, g, pa, pb);
                }
            }
            combine_rows(s_red, tr, tc, tpr, rows_par, pa, pb);
            if (tr == 0) {
                const bool chunk_end = (k + 1) % chunk == 0 || k + 1 == T, first = k < chunk;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    double* acc = s_acc + (4 * tc + i) * 2;
                    double* tot = s_tot + (4 * tc + i) * 2;
                    const double ca = acc[0] + pa[i], cb = acc[1] + pb[i];
                    if (chunk_end) {                                  // chunk 0 starts the total, the others are added to it
                        tot[0] = first ? ca : tot[0] + ca; tot[1] = first ? cb : tot[1] + cb;
                        acc[0] = 0; acc[1] = 0;
                    } else { acc[0] = ca; acc[1] = cb; }
                }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) {                                           // the partials of this slice have been read: leave; the last one resets
        const unsigned old = __hip_atomic_fetch_add(depart, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)T - 1u) {
            __hip_atomic_store(arrive, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(depart, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    // ---- per-channel results (every workgroup of the slice computes the same bits; tile 0 writes them out)
    if (threadIdx.x < 4 * s) {
        const int ch = slice * 4 * s + threadIdx.x;
        const double sa = s_tot[2 * threadIdx.x], sb = s_tot[2 * threadIdx.x + 1];
        if (!BWD) {
            const double n = (double)P.R, mean = sa / n;
            double var = sb / n - mean * mean; if (var < 0) var = 0;              // biased (normalisation)
            const float m = (float)mean, is = (float)(1.0 / sqrt(var + (double)P.eps));
            s_stat[2 * threadIdx.x] = m; s_stat[2 * threadIdx.x + 1] = is;
            if (tile == 0) {
                P.mean[ch] = m; P.invstd[ch] = is;
                if (ch == 0 && P.num_batches_tracked) P.num_batches_tracked[0] += 1;
                if (P.running_mean) {                                             // nn.BatchNorm2d: unbiased var into the running buffer
                    const double unb = n > 1 ? var * n / (n - 1) : var;
                    P.running_mean[ch] = (float)((1.0 - P.momentum) * P.running_mean[ch] + P.momentum * mean);
                    P.running_var[ch] = (float)((1.0 - P.momentum) * P.running_var[ch] + P.momentum * unb);
                }
            }
        } else {
            const float db = (float)sa, dg = (float)sb;
            s_stat[2 * threadIdx.x] = db; s_stat[2 * threadIdx.x + 1] = dg;
            if (tile == 0) { P.dbeta[ch] = db; P.dgamma[ch] = dg; }
        }
    }
    __syncthreads();

    stamp(5);
    // ---- phase 2: apply to the tile still held in registers
    const float* st = s_stat + 8 * tc;                                // (v0, w0, v1, w1, v2, w2, v3, w3) of this lane's channel quad
    const long long r0 = (long long)tile * P.rows_per_blk;
    // uniform tile bases + 32-bit lane offsets (a tile spans < 2^31 bytes: rows x row stride), advanced row lane by row lane
    const unsigned ostride = (unsigned)(BWD ? c4n : P.y_rs4) * QB, rstride = (unsigned)c4n * QB;
    char* obase = (BWD ? reinterpret_cast<char*>(P.gx) : reinterpret_cast<char*>(P.y)) + (size_t)r0 * ostride;
    const char* rbase = reinterpret_cast<const char*>(P.res) + (size_t)r0 * rstride;
    unsigned ooff = (unsigned)tr * ostride + (unsigned)q * QB;
    const unsigned ostep = (unsigned)rows_par * ostride;
    if (!BWD) {
        const float4 mu = make_float4(st[0], st[2], st[4], st[6]), is = make_float4(st[1], st[3], st[5], st[7]);
        const float4 ga = reinterpret_cast<const float4*>(P.gamma)[q], be = reinterpret_cast<const float4*>(P.beta)[q];
        const bool has_res = P.res != nullptr;
#pragma unroll
        for (int u0 = 0; u0 < KR; u0 += 4) {                          // the residual (ResBlock skip) arrives four rows at a time
        float4 rs[4];
        if (has_res) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                int r = tr + (u0 + u) * rows_par; if (r > my_rows - 1) r = my_rows - 1;
                rs[u] = ld_quad_at<BF>(rbase, (unsigned)r * rstride + (unsigned)q * QB);
            }
        }
#pragma unroll
        for (int u = u0; u < u0 + 4; ++u) {
            const int r = tr + u * rows_par;
            if (r < my_rows) {
                const float4 vv = hx[u];
                float4 o;
                if (has_res && P.res_pre) {                          // ResNet bottleneck: act(bn(x) + s)
                    const float4 tt = rs[u - u0];
                    o.x = act_fwd<ACT>((ga.x * ((vv.x - mu.x) * is.x) + be.x) + tt.x); o.y = act_fwd<ACT>((ga.y * ((vv.y - mu.y) * is.y) + be.y) + tt.y);
                    o.z = act_fwd<ACT>((ga.z * ((vv.z - mu.z) * is.z) + be.z) + tt.z); o.w = act_fwd<ACT>((ga.w * ((vv.w - mu.w) * is.w) + be.w) + tt.w);
                } else {
                    o.x = act_fwd<ACT>(ga.x * ((vv.x - mu.x) * is.x) + be.x); o.y = act_fwd<ACT>(ga.y * ((vv.y - mu.y) * is.y) + be.y);
                    o.z = act_fwd<ACT>(ga.z * ((vv.z - mu.z) * is.z) + be.z); o.w = act_fwd<ACT>(ga.w * ((vv.w - mu.w) * is.w) + be.w);
                    if (has_res) { const float4 tt = rs[u - u0]; o.x += tt.x; o.y += tt.y; o.z += tt.z; o.w += tt.w; }
                }
                st_quad_at<BF>(obase, ooff, o);
            }
            ooff += ostep;
            __builtin_amdgcn_sched_barrier(0);                        // row by row: the tile itself is the register budget
        }
        }
    } else {
        const float invR = (float)(1.0 / (double)P.R);
        const float4 db = make_float4(st[0], st[2], st[4], st[6]), dg = make_float4(st[1], st[3], st[5], st[7]);
#pragma unroll
        for (int u = 0; u < KR; ++u) {
            const int r = tr + u * rows_par;
            if (r < my_rows) {
                const float4 xh = hx[u], gz = hg[BWD ? u : 0];
                float4 o;
                o.x = c.ga.x * c.is.x * (gz.x - db.x * invR - xh.x * dg.x * invR); o.y = c.ga.y * c.is.y * (gz.y - db.y * invR - xh.y * dg.y * invR);
                o.z = c.ga.z * c.is.z * (gz.z - db.z * invR - xh.z * dg.z * invR); o.w = c.ga.w * c.is.w * (gz.w - db.w * invR - xh.w * dg.w * invR);
                st_quad_at<BF>(obase, ooff, o);
            }
            ooff += ostep;
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    stamp(6);
}

constexpr size_t kLds = (size_t)kT * 8 * sizeof(double) + 384 * sizeof(double) + 128 * sizeof(float) + 16;

struct Resident { int per_cu[2][2][4][2]; bool known[2][2][4][2]; int cus; unsigned* sync; bool ready; };
Resident g_dev[16];
std::atomic<unsigned> g_slot{0};
std::atomic<unsigned> g_launches{0};

template <bool BWD, bool BF, int ACT, int KR>
int launch_one(const BnParams& P, const BnFusedPlan& pl, Resident& D, hipStream_t stream) {
    constexpr int ki = KR == 8 ? 0 : 1;
    auto kern = bn_fused_kernel<BWD, BF, ACT, KR>;
    if (!D.known[BWD][BF][ACT][ki]) {
        int n = 0;
        ODPL_HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, kT, kLds));
        D.per_cu[BWD][BF][ACT][ki] = n; D.known[BWD][BF][ACT][ki] = true;
    }
    // the occupancy query can read one workgroup per CU high where the scalar registers, not the vector ones, bound it (7-8 per
    // CU: MI355X_MICROARCH.md, residency): keep one back there.  A wrong answer costs time (bounded wait, self-service), not progress.
    const int api = D.per_cu[BWD][BF][ACT][ki], per_cu = api >= 7 ? api - 1 : api;
    if (per_cu < 1 || (long long)pl.T * pl.nslices > (long long)per_cu * D.cus) return 1;
    hipLaunchKernelGGL(kern, dim3(pl.T, pl.nslices), dim3(kT), kLds, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    g_launches.fetch_add(1);
    return 0;
}

template <bool BWD, bool BF, int ACT>
int launch_kr(const BnParams& P, const BnFusedPlan& pl, Resident& D, hipStream_t stream) {
    if (pl.kr == 8) return launch_one<BWD, BF, ACT, 8>(P, pl, D, stream);
    if (BWD) return 1;                                                // 16 rows of (xhat, gz) per lane do not fit the register budget
    return launch_one<false, BF, ACT, 16>(P, pl, D, stream);
}
template <bool BWD, bool BF>
int launch_act(const BnParams& P, const BnFusedPlan& pl, Resident& D, hipStream_t stream) {
    switch (P.act) {
        case 0: return launch_kr<BWD, BF, 0>(P, pl, D, stream);
        case 1: return launch_kr<BWD, BF, 1>(P, pl, D, stream);
        case 2: return launch_kr<BWD, BF, 2>(P, pl, D, stream);
        default: return launch_kr<BWD, BF, 3>(P, pl, D, stream);
    }
}
}  // namespace

namespace odpl {

// Geometry: s quads per slice (lanes along the row), KR rows per lane, T row tiles per slice.  Workgroups = bytes / (32 KB * KR / 8);
// what a workgroup reads back in phase 2 = T * 4 s * 16 B.  The rules below pick, per map: 16-quad slices where T stays small, 8-quad
// slices (128-byte row segments: still whole cache lines) for the longer maps, 16 rows per lane where 8 would need > 128 tiles.
bool bn_fused_plan(long long R, int C, bool bf16, bool bwd, BnFusedPlan* p) {
    (void)bf16;
    if (!g_bn_fused || R < 1 || C < 32 || (C & 3)) return false;
    const int c4n = C >> 2;
    const long long bytes = R * C * 4;                                  // register footprint is fp32 either way
    if (bytes * (bwd ? 2 : 1) > (96LL << 20)) return false;             // what 256 CUs hold with a margin (forward <= 96 MB, backward <= 48 MB)
    int best_s = 0, best_kr = 0; long long best_T = 0;
    for (int kr = 8; kr <= (bwd ? 8 : 16) && !best_s; kr *= 2)
        for (int s = 16; s >= 8; s >>= 1) {
            if (c4n % s) continue;
            const long long rpb = (long long)kr * (kT / s), T = (R + rpb - 1) / rpb;
            if (T > 128 || T * 4 * s * 16 > (48 << 10)) continue;
            if ((T + (kT / (8 * s)) - 1) / (kT / (8 * s)) > kMaxChunk) continue;      // tiles per reduction thread
            if (c4n / s > kMaxSlices) continue;
            best_s = s; best_kr = kr; best_T = T;
            break;
        }
    if (!best_s) return false;
    p->s = best_s; p->kr = best_kr; p->T = (int)best_T; p->nslices = c4n / best_s; p->rpb = best_kr * (kT / best_s); p->lds = kLds;
    return true;
}

size_t bn_fused_workspace_bytes(long long R, int C) {
    size_t need = 0;
    for (int bwd = 0; bwd < 2; ++bwd) {
        BnFusedPlan pl;
        if (bn_fused_plan(R, C, false, bwd != 0, &pl)) { const size_t n = (size_t)pl.T * C * 2 * sizeof(double); if (n > need) need = n; }
    }
    return need;
}

int bn_fused_launch(const BnParams& P0, const BnFusedPlan& pl, bool bwd, bool bf16, hipStream_t stream) {
    int dev = 0;
    ODPL_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) return 1;
    Resident& D = g_dev[dev];
    if (!D.ready) {
        hipDeviceProp_t prop;
        ODPL_HIP_TRY(hipGetDeviceProperties(&prop, dev));
        D.cus = prop.multiProcessorCount;
        ODPL_HIP_TRY(hipGetSymbolAddress((void**)&D.sync, HIP_SYMBOL(g_bn_sync)));
        D.ready = true;
    }
    BnParams P = P0;
    P.nblk = pl.T; P.rows_per_blk = pl.rpb; P.slice_q = pl.s;
    P.sync = D.sync + (size_t)(g_slot.fetch_add(1) % kSlots) * kMaxSlices * 2;
    P.timeout_ticks = (unsigned)(g_bn_fused_timeout_us < 0 ? 0 : g_bn_fused_timeout_us) * 100u;
    P.stamp = nullptr;
    if (g_stamp_on) ODPL_HIP_TRY(hipGetSymbolAddress((void**)&P.stamp, HIP_SYMBOL(g_bn_stamp)));
    if (bwd) return bf16 ? launch_act<true, true>(P, pl, D, stream) : launch_act<true, false>(P, pl, D, stream);
    return bf16 ? launch_act<false, true>(P, pl, D, stream) : launch_act<false, false>(P, pl, D, stream);
}

}  // namespace odpl

// diagnostic: on != 0 -> later one-launch BatchNorm launches record 8 readings of the 100 MHz clock per workgroup (start, sums done,
// partials drained, peers seen, acquired, statistics done, stores issued); out (host, n_wg * 8 u64) receives the last launch's
extern "C" int odpl_bn_fused_debug_stamps(int32_t on, uint64_t* out, int32_t n_wg) {
    g_stamp_on = on != 0;
    if (out && n_wg > 0) {
        if (n_wg > kStampWgs) n_wg = kStampWgs;
        ODPL_HIP_TRY(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bn_stamp), (size_t)n_wg * 8 * sizeof(uint64_t)));
    }
    return 0;
}

// diagnostic: out[0] = one-launch BatchNorm launches since the library was loaded, out[1] = workgroups among them that did not wait
// for their slice (self-service path); synchronises the device
extern "C" int odpl_bn_fused_stats(int32_t* out) {
    if (!out) return ODPL_EINVAL;
    unsigned v = 0;
    ODPL_HIP_TRY(hipMemcpyFromSymbol(&v, HIP_SYMBOL(g_bn_fallbacks), sizeof(v)));
    out[0] = (int32_t)g_launches.load();
    out[1] = (int32_t)v;
    return 0;
}
