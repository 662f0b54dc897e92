// HBM write/read bandwidth probe: what can a plain streaming kernel reach on this chip at a given size?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
template <bool NT> __global__ void fill(v4f* p, size_t n4) {
    const v4f z = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (NT) __builtin_nontemporal_store(z, p + i); else p[i] = z;
    }
}
template <bool NT> __global__ void fill_chunk(v4f* p, size_t n4, int per_block) {
    const v4f z = {0, 0, 0, 0};
    size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block; if (hi > n4) hi = n4;
    for (size_t i = lo + threadIdx.x; i < hi; i += blockDim.x) { if (NT) __builtin_nontemporal_store(z, p + i); else p[i] = z; }
}
__global__ void readk(const v4f* p, size_t n4, float* out) {
    v4f a = {0, 0, 0, 0};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
int main() {
    const size_t big = 1ull << 30;
    float *buf, *evict, *out;
    hipMalloc(&buf, big); hipMalloc(&evict, big); hipMalloc(&out, 16);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t sizes[] = {58ull << 20, 128ull << 20, 270ull << 20, 512ull << 20, 1024ull << 20};
    printf("kind,MB,blocks,us,GBps\n");
    for (size_t sz : sizes) {
        size_t n4 = sz / 16;
        for (int variant = 0; variant < 6; ++variant) {
            float best = 1e9;
            int blocks = 0;
            for (int rep = 0; rep < 6; ++rep) {
                hipMemsetAsync(evict, 1, big, 0);            // evict L2 + Infinity Cache
                hipEventRecord(a, 0);
                switch (variant) {
                    case 0: blocks = 2048; fill<false><<<blocks, 256>>>((v4f*)buf, n4); break;
                    case 1: blocks = 2048; fill<true><<<blocks, 256>>>((v4f*)buf, n4); break;
                    case 2: blocks = (int)((n4 + 5119) / 5120); fill_chunk<true><<<blocks, 256>>>((v4f*)buf, n4, 5120); break;
                    case 3: blocks = (int)((n4 + 2047) / 2048); fill_chunk<true><<<blocks, 256>>>((v4f*)buf, n4, 2048); break;
                    case 4: blocks = 2048; readk<<<blocks, 256>>>((const v4f*)buf, n4, out); break;
                    case 5: blocks = 8192; fill<true><<<blocks, 256>>>((v4f*)buf, n4); break;
                }
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            const char* names[] = {"fill_stride", "fill_stride_nt", "fill_chunk5120_nt", "fill_chunk2048_nt", "read_stride", "fill_stride_nt_8192"};
            printf("%s,%zu,%d,%.1f,%.0f\n", names[variant], sz >> 20, blocks, best * 1e3, sz / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
