// Which structure reads a [rows, 80] fp32 matrix fastest when every row needs a reduction over its 320 contiguous bytes?
// (greedy_nms_prep_kernel's access pattern: 4 x 200 700 rows = 257 MB at BASELINE config 4.)  Variants:
//   A  workgroup tile of 64 rows staged in LDS, loads e = tid + u*256 (5 in flight), barrier, 4 lanes per row   [the shipped structure]
//   B  same grid, no LDS: lanes just fold what they load (upper bound for this grid shape)
//   C  wave-local tiles: each wave stages its own 16 rows (5 KB contiguous) in its LDS slice, no workgroup barrier
//   D  A with a wave-contiguous load order (wave w reads the 5 KB run [w*320, (w+1)*320) of the tile)
//   E  C with 32 rows per wave (10 loads in flight per lane)
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 tools/probe_rowmax.hip -o gpurun_out/probe_rowmax && gpurun_out/probe_rowmax
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
constexpr int C = 80, C4 = 20, LD = 84;

__device__ __forceinline__ void row_reduce(const float* row, int q, float& m, int& bi) {
    m = -INFINITY; bi = 0x7fffffff;
    for (int k = q; k < C; k += 4) { const float v = row[k]; if (v > m) { m = v; bi = k; } }
    for (int o = 1; o < 4; o <<= 1) {
        const float om = __shfl_xor(m, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if (om > m || (om == m && oi < bi)) { m = om; bi = oi; }
    }
}

__global__ __launch_bounds__(256) void varA(const float* src, int rows, float* score, int* label) {
    __shared__ __align__(16) float s[64 * LD];
    const int tid = threadIdx.x, a0 = blockIdx.x * 64, nr = min(64, rows - a0), n4 = nr * C4;
    const float4* p = reinterpret_cast<const float4*>(src + (size_t)a0 * C);
    float4 v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) v[u] = tid + u * 256 < n4 ? p[tid + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = tid + u * 256; if (e < n4) { const int r = e / C4; *reinterpret_cast<float4*>(&s[r * LD + 4 * (e - r * C4)]) = v[u]; } }
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    if (r < nr) { float m; int bi; row_reduce(s + r * LD, q, m, bi); if (q == 0) { score[a0 + r] = m; label[a0 + r] = bi; } }
}
__global__ __launch_bounds__(256) void varB(const float* src, int rows, float* score, int* label) {
    const int tid = threadIdx.x, a0 = blockIdx.x * 64, nr = min(64, rows - a0), n4 = nr * C4;
    const float4* p = reinterpret_cast<const float4*>(src + (size_t)a0 * C);
    float m = -INFINITY;
#pragma unroll
    for (int u = 0; u < 5; ++u) { if (tid + u * 256 < n4) { const float4 v = p[tid + u * 256]; m = fmaxf(fmaxf(m, fmaxf(v.x, v.y)), fmaxf(v.z, v.w)); } }
    if (m == 12345.f) { score[a0] = m; label[a0] = 1; }
}
template <int RW>   // rows per wave
__global__ __launch_bounds__(256) void varC(const float* src, int rows, float* score, int* label) {
    __shared__ __align__(16) float s[4 * RW * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int a0 = (blockIdx.x * 4 + w) * RW, nr = min(RW, rows - a0), n4 = nr * C4;
    if (nr <= 0) return;
    float* sw = s + w * RW * LD;
    const float4* p = reinterpret_cast<const float4*>(src + (size_t)a0 * C);
    constexpr int U = RW * C4 / 64;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = lane + u * 64 < n4 ? p[lane + u * 64] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < U; ++u) { const int e = lane + u * 64; if (e < n4) { const int r = e / C4; *reinterpret_cast<float4*>(&sw[r * LD + 4 * (e - r * C4)]) = v[u]; } }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int r0 = 0; r0 < RW; r0 += 16) {
        const int r = r0 + (lane >> 2), q = lane & 3;
        if (r < nr) { float m; int bi; row_reduce(sw + r * LD, q, m, bi); if (q == 0) { score[a0 + r] = m; label[a0 + r] = bi; } }
    }
}
__global__ __launch_bounds__(256) void varD(const float* src, int rows, float* score, int* label) {
    __shared__ __align__(16) float s[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, a0 = blockIdx.x * 64, nr = min(64, rows - a0), n4 = nr * C4;
    const float4* p = reinterpret_cast<const float4*>(src + (size_t)a0 * C);
    float4 v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = w * 320 + u * 64 + lane; v[u] = e < n4 ? p[e] : make_float4(0, 0, 0, 0); }
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = w * 320 + u * 64 + lane; if (e < n4) { const int r = e / C4; *reinterpret_cast<float4*>(&s[r * LD + 4 * (e - r * C4)]) = v[u]; } }
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    if (r < nr) { float m; int bi; row_reduce(s + r * LD, q, m, bi); if (q == 0) { score[a0 + r] = m; label[a0 + r] = bi; } }
}

// F: the shipped kernel's arithmetic (sigmoid only on possible ties, NaN-aware compare, tile counts) on variant A's structure
__global__ __launch_bounds__(256) void varF(const float* src, int rows, float* score, int* label, int* tile_cnt) {
    __shared__ __align__(16) float s[64 * LD];
    __shared__ int s_tc[4];
    const int tid = threadIdx.x, a0 = blockIdx.x * 64, nr = min(64, rows - a0), n4 = nr * C4;
    const float4* p = reinterpret_cast<const float4*>(src + (size_t)a0 * C);
    float4 v[5];
#pragma unroll
    for (int u = 0; u < 5; ++u) v[u] = tid + u * 256 < n4 ? p[tid + u * 256] : make_float4(0, 0, 0, 0);
#pragma unroll
    for (int u = 0; u < 5; ++u) { const int e = tid + u * 256; if (e < n4) { const int r = e / C4; *reinterpret_cast<float4*>(&s[r * LD + 4 * (e - r * C4)]) = v[u]; } }
    __syncthreads();
    const int r = tid >> 2, q = tid & 3;
    float best = -1.f; int bi = 0x7fffffff;
    if (r < nr) {
        const float* row = s + r * LD;
        float m = -INFINITY;
        for (int k = q; k < C; k += 4) m = fmaxf(m, row[k]);
        m = fmaxf(m, __shfl_xor(m, 1, 64)); m = fmaxf(m, __shfl_xor(m, 2, 64));
        const float thr = m - 0.0625f; const bool all = !(m >= -80.0f);
        unsigned long long need = 0ull; int j = 0;
        for (int k = q; k < C; k += 4, ++j) { const float l = row[k]; if (all || !(l < thr) || l > 8.0f) need |= 1ull << j; }
        while (need) {
            const int k = q + 4 * __builtin_ctzll(need); need &= need - 1ull;
            const float sg = 1.0f / (1.0f + expf(-row[k]));
            if (sg > best || (sg != sg && best == best)) { best = sg; bi = k; }
        }
    }
    for (int o = 1; o < 4; o <<= 1) {
        const float ob = __shfl_xor(best, o, 64); const int oi = __shfl_xor(bi, o, 64);
        if ((ob != ob && (best == best || oi < bi)) || (best == best && (ob > best || (ob == best && oi < bi)))) { best = ob; bi = oi; }
    }
    if (r < nr && q == 0) { score[a0 + r] = best; label[a0 + r] = bi; }
    const unsigned long long pm = __ballot(r < nr && q == 0 && best > 0.45f);
    if ((tid & 63) == 0) s_tc[tid >> 6] = __popcll(pm);
    __syncthreads();
    if (tid == 0) tile_cnt[blockIdx.x] = s_tc[0] + s_tc[1] + s_tc[2] + s_tc[3];
}
__global__ void fill_random(float* p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(h & 0xffffff) * (12.0f / 16777216.0f) - 9.0f;               // logits in [-9, 3)
    }
}

int main() {
    const int rows = 4 * 200700;
    const size_t bytes = (size_t)rows * C * 4, big = 1ull << 30;
    float *src, *evict, *score; int* label;
    hipMalloc(&src, bytes); hipMalloc(&evict, big); hipMalloc(&score, rows * 4); hipMalloc(&label, rows * 4);
    hipMemset(src, 0, bytes);
    int* tile_cnt; hipMalloc(&tile_cnt, ((rows + 63) / 64) * 4);
    float* sink; hipMalloc(&sink, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const char* names[] = {"A_wg64_lds", "B_noLDS_bound", "C_wave16_lds", "D_wg64_wavecontig", "E_wave32_lds", "F_shipped_math_zeros",
                           "A_random_data", "B_random_data", "F_shipped_math_random"};
    printf("variant,us,GBps\n");
    for (int vi = 0; vi < 9; ++vi) {
        int variant = vi;
        if (vi == 6) { hipLaunchKernelGGL(fill_random, dim3(4096), dim3(256), 0, 0, src, bytes / 4); variant = 0; }
        if (vi == 7) variant = 1;
        if (vi == 8) variant = 5;
        float best = 1e9, med[9];
        for (int rep = 0; rep < 9; ++rep) {
            hipMemsetAsync(evict, 1, big, 0);                       // dirty eviction would halve reads: follow with a clean pass
            hipMemcpyAsync(sink, evict, 16, hipMemcpyDeviceToDevice, 0);
            // clean eviction: read 1 GiB
            hipLaunchKernelGGL(varB, dim3((unsigned)(big / 4 / C / 64)), dim3(256), 0, 0, evict, (int)(big / 4 / C), score, label);
            hipEventRecord(a, 0);
            const int wg64 = (rows + 63) / 64;
            switch (variant) {
                case 0: hipLaunchKernelGGL(varA, dim3(wg64), dim3(256), 0, 0, src, rows, score, label); break;
                case 1: hipLaunchKernelGGL(varB, dim3(wg64), dim3(256), 0, 0, src, rows, score, label); break;
                case 2: hipLaunchKernelGGL(varC<16>, dim3(wg64), dim3(256), 0, 0, src, rows, score, label); break;
                case 3: hipLaunchKernelGGL(varD, dim3(wg64), dim3(256), 0, 0, src, rows, score, label); break;
                case 4: hipLaunchKernelGGL(varC<32>, dim3((rows + 127) / 128), dim3(256), 0, 0, src, rows, score, label); break;
                case 5: hipLaunchKernelGGL(varF, dim3(wg64), dim3(256), 0, 0, src, rows, score, label, tile_cnt); break;
            }
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            med[rep] = ms; if (ms < best) best = ms;
        }
        for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) if (med[j] < med[i]) { float t = med[i]; med[i] = med[j]; med[j] = t; }
        printf("%s,%.1f,%.0f\n", names[vi], med[4] * 1e3, bytes / (med[4] * 1e-3) / 1e9);
    }
    return 0;
}
