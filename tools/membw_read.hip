// HBM read probe: how fast can a cold 58-270 MB buffer be read, by load width / loads in flight / grid size?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int U> __global__ void read_u(const v4f* p, size_t n4, float* out) {
    v4f a = {0, 0, 0, 0};
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u];
    }
    for (; i < n4; i += stride) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
template <int U> __global__ void read_chunk(const v4f* p, size_t n4, int per_block, float* out) {
    v4f a = {0, 0, 0, 0};
    size_t lo = (size_t)blockIdx.x * per_block, hi = lo + per_block; if (hi > n4) hi = n4;
    size_t i = lo + threadIdx.x;
    for (; i + (U - 1) * blockDim.x < hi; i += U * blockDim.x) {
        v4f v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = p[i + u * blockDim.x];
#pragma unroll
        for (int u = 0; u < U; ++u) a += v[u];
    }
    for (; i < hi; i += blockDim.x) a += p[i];
    if (a.x + a.y + a.z + a.w == 12345.f) out[0] = 1;
}
__global__ void read_dword(const float* p, size_t n, float* out) {
    float a = 0;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) a += v[u];
    }
    if (a == 12345.f) out[0] = 1;
}
int main() {
    const size_t big = 1ull << 30;
    float *buf, *evict, *out;
    hipMalloc(&buf, big); hipMalloc(&evict, big); hipMalloc(&out, 16);
    hipMemset(buf, 0, big);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    size_t sizes[] = {58ull << 20, 128ull << 20, 270ull << 20};
    printf("kind,MB,blocks,us,GBps\n");
    for (size_t sz : sizes) {
        size_t n4 = sz / 16;
        for (int variant = 0; variant < 8; ++variant) {
            float best = 1e9; int blocks = 0; const char* name = "";
            for (int rep = 0; rep < 6; ++rep) {
                read_u<4><<<4096, 256>>>((const v4f*)evict, big / 16, out);   // evict with CLEAN lines (a write-evict leaves
                                                                               // 256 MB of dirty lines draining under the timed reads)
                hipEventRecord(a, 0);
                switch (variant) {
                    case 0: name = "f4_stride_u1_2048"; blocks = 2048; read_u<1><<<blocks, 256>>>((const v4f*)buf, n4, out); break;
                    case 1: name = "f4_stride_u4_2048"; blocks = 2048; read_u<4><<<blocks, 256>>>((const v4f*)buf, n4, out); break;
                    case 2: name = "f4_stride_u8_2048"; blocks = 2048; read_u<8><<<blocks, 256>>>((const v4f*)buf, n4, out); break;
                    case 3: name = "f4_stride_u4_8192"; blocks = 8192; read_u<4><<<blocks, 256>>>((const v4f*)buf, n4, out); break;
                    case 4: name = "f4_chunk2048_u4"; blocks = (int)((n4 + 2047) / 2048); read_chunk<4><<<blocks, 256>>>((const v4f*)buf, n4, 2048, out); break;
                    case 5: name = "f4_chunk4096_u8"; blocks = (int)((n4 + 4095) / 4096); read_chunk<8><<<blocks, 256>>>((const v4f*)buf, n4, 4096, out); break;
                    case 6: name = "f4_chunk1024_u4"; blocks = (int)((n4 + 1023) / 1024); read_chunk<4><<<blocks, 256>>>((const v4f*)buf, n4, 1024, out); break;
                    case 7: name = "dword_stride_u8_4096"; blocks = 4096; read_dword<<<blocks, 256>>>(buf, sz / 4, out); break;
                }
                hipEventRecord(b, 0); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b);
                if (ms < best) best = ms;
            }
            printf("%s,%zu,%d,%.1f,%.0f\n", name, sz >> 20, blocks, best * 1e3, sz / (best * 1e-3) / 1e9);
        }
    }
    return 0;
}
