// What does a plain copy reach on this chip at the size of the head-gradient stream (77 MB in, 77 MB out), cold?
// Workgroup = 256 lanes x 8 16-B vectors (32 KB in, 32 KB out), all loads issued before the first store -- the access shape of
// headgrad_stream_kernel without its arithmetic, LDS tile and partial sums.  Eviction: 1 GiB is READ before every timed launch.
// hipcc --offload-arch=gfx950 -O3 tools/probe_copy.hip -o gpurun_out/probe_copy && gpurun_out/probe_copy
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool NTS = NT>
__global__ __launch_bounds__(256) void copyk(const v4f* __restrict__ src, v4f* __restrict__ dst, size_t n4) {
    const size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    v4f v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; v[u] = i < n4 ? (NT ? __builtin_nontemporal_load(src + i) : src[i]) : v4f{0, 0, 0, 0}; }
#pragma unroll
    for (int u = 0; u < U; ++u) { const size_t i = base + (size_t)u * 256; if (i < n4) { if (NTS) __builtin_nontemporal_store(v[u], dst + i); else dst[i] = v[u]; } }
}
__global__ void readk(const v4f* p, size_t n4, float* out) {
    v4f a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
int main() {
    const size_t big = 1ull << 30;
    float *src, *dst, *evict, *out;
    (void)hipMalloc(&src, 512ull << 20); (void)hipMalloc(&dst, 512ull << 20); (void)hipMalloc(&evict, big); (void)hipMalloc(&out, 16);
    (void)hipMemset(src, 0, 512ull << 20); (void)hipMemset(evict, 0, big);
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const size_t sizes[] = {77ull * 1000 * 1000, 270ull << 20};
    printf("kind,MB_each_way,us,GBps_read_plus_write\n");
    for (size_t sz : sizes) {
        const size_t n4 = sz / 16;
        for (int variant = 0; variant < 6; ++variant) {
            float t[9];
            for (int rep = 0; rep < 9; ++rep) {
                hipLaunchKernelGGL(readk, dim3(4096), dim3(256), 0, 0, (const v4f*)evict, big / 16, out);
                (void)hipEventRecord(a, 0);
                switch (variant) {
                    case 0: hipLaunchKernelGGL((copyk<8, true>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                    case 1: hipLaunchKernelGGL((copyk<8, false>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                    case 2: hipLaunchKernelGGL((copyk<4, true>), dim3((unsigned)((n4 + 1023) / 1024)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                    case 4: hipLaunchKernelGGL((copyk<8, true, false>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                    case 5: hipLaunchKernelGGL((copyk<8, false, true>), dim3((unsigned)((n4 + 2047) / 2048)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                    case 3: hipLaunchKernelGGL((copyk<1, true>), dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, 0, (const v4f*)src, (v4f*)dst, n4); break;
                }
                (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
                (void)hipEventElapsedTime(&t[rep], a, b);
            }
            for (int i = 0; i < 9; ++i) for (int j = i + 1; j < 9; ++j) if (t[j] < t[i]) { float x = t[i]; t[i] = t[j]; t[j] = x; }
            const char* names[] = {"copy_u8_nt", "copy_u8", "copy_u4_nt", "copy_u1_nt", "copy_u8_ntload_only", "copy_u8_ntstore_only"};
            printf("%s,%.0f,%.1f,%.0f\n", names[variant], sz / 1e6, t[4] * 1e3, 2.0 * sz / (t[4] * 1e-3) / 1e9);
        }
    }
    return 0;
}
