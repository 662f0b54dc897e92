// Shared device helpers for libodpl_hip (gfx950 only; wave = 64 lanes).
// Built with -ffp-contract=off: index-feeding arithmetic (anchor IoU, cell indices) must round
// exactly like the reference's separately-rounded fp32 torch ops.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define ODPL_WAVE 64

#define ODPL_HIP_TRY(expr)                        \
    do {                                          \
        hipError_t _e = (expr);                   \
        if (_e != hipSuccess) return (int)_e;     \
    } while (0)

namespace odpl {

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o, 64));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
    return v;
}

// Block-wide sum of N floats per thread; result valid in thread 0. `sh` holds N*(blockDim/64) floats.
// Fixed shuffle tree + fixed wave order => bitwise reproducible run to run.
template <int N>
__device__ __forceinline__ void block_sum(float (&v)[N], float* sh) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i) v[i] = wave_sum(v[i]);
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) sh[wv * N + i] = v[i];
    }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            float a = sh[i];
            for (int w = 1; w < nw; ++w) a += sh[w * N + i];
            v[i] = a;
        }
    }
    __syncthreads();
}

__device__ __forceinline__ float sigmoidf_(float z) { return 1.0f / (1.0f + expf(-z)); }
// Hardware exp2 / reciprocal (1 ulp each; the scaled argument adds |z| * 2^-24 relative): for inference outputs whose
// tests carry a 1e-5 relative tolerance.  -inf -> 0, +inf -> 1, NaN -> NaN like the exact form.
__device__ __forceinline__ float sigmoid_fast(float z) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.44269504088896f)); }
// e / d for 0 <= e < 2^16 and 1 <= d <= 2^16 with the host-side magic number ceil(2^32 / d) (exact: the excess e*r/(d*2^32) stays
// below 1/d); d = 1 has no 32-bit magic and is passed as 0
__device__ __forceinline__ int fastdiv16(int e, unsigned magic) { return magic ? (int)__umulhi((unsigned)e, magic) : e; }
inline unsigned fastdiv16_magic(unsigned d) { return d <= 1 ? 0u : (unsigned)((0xffffffffull + d) / d); }

// Workgroup barrier that orders LDS traffic only.  __syncthreads() is a workgroup-scope fence as well, i.e. `s_waitcnt vmcnt(0)`:
// every X load and every dX store of the wave would have to retire before the barrier.  Use it where a barrier
// only publishes LDS data: the wave's global loads and stores then stay in flight across it.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// nn.BCELoss element (log clamped at -100; ATen uses log1p(-p) for the negative branch).
__device__ __forceinline__ float bce_elem(float p, float t) {
    return -(t * fmaxf(logf(p), -100.0f) + (1.0f - t) * fmaxf(log1pf(-p), -100.0f));
}
// d BCE / d logit through sigmoid, following autograd literally:
// binary_cross_entropy_backward = (p - t) / max((1-p)p, 1e-12) * g ; sigmoid_backward = * (1-p) * p
__device__ __forceinline__ float bce_grad_logit(float p, float t, float g) {
    const float gp = (p - t) / fmaxf((1.0f - p) * p, 1e-12f) * g;
    return gp * (1.0f - p) * p;
}

// coord criterion element + derivative: SmoothL1 (beta = 1) or MSE
__device__ __forceinline__ float coord_elem(float d, int smooth_l1) {
    if (smooth_l1) {
        const float a = fabsf(d);
        return a < 1.0f ? 0.5f * d * d : a - 0.5f;
    }
    return d * d;
}
__device__ __forceinline__ float coord_deriv(float d, int smooth_l1) {
    if (smooth_l1) return fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f));
    return 2.0f * d;
}

typedef float v4f __attribute__((ext_vector_type(4)));

// Zero base[lo, hi) with the whole workgroup (THREADS threads): 16-B stores on the aligned middle,
// <=3 scalar stores at each end.  base must be 16-byte aligned; lo/hi are element offsets.
// NT: non-temporal stores (streaming zeros that nobody in this kernel re-reads).
template <bool NT, int THREADS>
__device__ __forceinline__ void fill_zero(float* __restrict__ base, size_t lo, size_t hi) {
    const int tid = threadIdx.x;
    size_t lo4 = (lo + 3) & ~(size_t)3; if (lo4 > hi) lo4 = hi;
    size_t hi4 = hi & ~(size_t)3; if (hi4 < lo4) hi4 = lo4;
    if ((size_t)tid < lo4 - lo) base[lo + tid] = 0.f;
    if ((size_t)tid < hi - hi4) base[hi4 + tid] = 0.f;
    v4f* p4 = reinterpret_cast<v4f*>(base);
    const v4f z = {0.f, 0.f, 0.f, 0.f};
    size_t i = lo4 / 4 + tid;
    const size_t e = hi4 / 4;
    for (; i + 3 * THREADS < e; i += 4 * THREADS) {      // 4 independent 1-KiB-per-wave stores in flight
        if (NT) {
            __builtin_nontemporal_store(z, p4 + i);
            __builtin_nontemporal_store(z, p4 + i + THREADS);
            __builtin_nontemporal_store(z, p4 + i + 2 * THREADS);
            __builtin_nontemporal_store(z, p4 + i + 3 * THREADS);
        } else {
            p4[i] = z; p4[i + THREADS] = z; p4[i + 2 * THREADS] = z; p4[i + 3 * THREADS] = z;
        }
    }
    for (; i < e; i += THREADS) {
        if (NT) __builtin_nontemporal_store(z, p4 + i); else p4[i] = z;
    }
}

// torch `.long()` of an fp32 value: truncate toward zero.
__device__ __forceinline__ long long trunc_ll(float v) { return (long long)v; }

// Activations are fp32 (a channel quad = one 16-B float4) or bf16 (a quad = 8 B): statistics, affine parameters and all
// arithmetic are fp32 either way; bf16 results are rounded once (round-to-nearest-even), after the activation.
template <bool BF>
__device__ __forceinline__ float4 ld_quad(const void* p, long long qi) {
    if (!BF) return reinterpret_cast<const float4*>(p)[qi];
    const uint2 u = reinterpret_cast<const uint2*>(p)[qi];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}
__device__ __forceinline__ unsigned bf16_rne(float f) {
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;          // NaN stays NaN (quiet)
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
template <bool BF>
__device__ __forceinline__ void st_quad(void* p, long long qi, const float4& v) {
    if (!BF) { reinterpret_cast<float4*>(p)[qi] = v; return; }
    uint2 u;
    u.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16); u.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
    reinterpret_cast<uint2*>(p)[qi] = u;
}


// Byte fill as a KERNEL.  hipMemsetAsync is avoided on purpose: captured into a HIP graph it becomes a memset node that does
// not replay correctly on this stack (ROCm 7.0: state maps came back dirty on every replay after the first -- found by
// tests/test_gpu_graph.py); a kernel node replays like any other launch, so every entry point stays graph-capturable.
static __global__ void fill_bytes_kernel(unsigned char* __restrict__ p, unsigned v32, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t head = ((16 - ((uintptr_t)p & 15)) & 15) < n ? ((16 - ((uintptr_t)p & 15)) & 15) : n;    // bytes up to 16-B alignment
    const size_t n16 = (n - head) / 16;
    uint4* q = reinterpret_cast<uint4*>(p + head);
    const uint4 v = make_uint4(v32, v32, v32, v32);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) q[i] = v;
    const size_t tail0 = head + n16 * 16;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < head + (n - tail0); i += stride) {
        const size_t j = i < head ? i : tail0 + (i - head);
        p[j] = (unsigned char)v32;
    }
}
inline hipError_t fill_bytes(void* p, int byte_value, size_t n, hipStream_t stream) {
    if (n == 0) return hipSuccess;
    const unsigned b = (unsigned)byte_value & 0xffu, v32 = b | (b << 8) | (b << 16) | (b << 24);
    size_t blocks = (n / 16 + 255) / 256; if (blocks < 1) blocks = 1; if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(fill_bytes_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (unsigned char*)p, v32, n);
    return hipGetLastError();
}

}  // namespace odpl
