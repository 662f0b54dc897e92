// Does a second pass over a map hit the XCD-local L2 when its workgroups touch the same 4-KB chunks as the first pass's workgroups of
// the same XCD?  (BN: statistics pass then apply pass over x; backward: reduce pass then dx pass over dy and x.)
//   pass A  : sum of the buffer (a read stream), grid-stride over 4-KB chunks, block b takes chunks b, b + G, ...
//   pass B  : y = 2 * x (read + write), same grid; `shift` = 0: the same chunk -> block mapping as A (same XCD: blocks are dealt to
//             XCDs round robin), shift = 1: block b takes A's chunks of block b + 1 (another XCD); `reverse` walks the trips backwards
// Prints one line per (size, shift, reverse): the average time of pass B when it follows pass A.
// build: hipcc --offload-arch=gfx950 -O3 -o l2_reuse l2_reuse.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(256) void pass_a(const float4* x, long long n4, float* out) {
    const long long stride = (long long)gridDim.x * 256;
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) { const float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 12345.678f) out[0] = s;
}
__global__ __launch_bounds__(256) void pass_b(const float4* x, float4* y, long long n4, int shift, int reverse) {
    const long long stride = (long long)gridDim.x * 256;
    const long long b = (blockIdx.x + shift) % gridDim.x;
    const long long first = b * 256 + threadIdx.x;
    if (first >= n4) return;
    const long long trips = (n4 - first + stride - 1) / stride;
    for (long long k = 0; k < trips; ++k) {
        const long long i = first + (reverse ? trips - 1 - k : k) * stride;
        const float4 v = x[i];
        y[i] = make_float4(2.f * v.x, 2.f * v.y, 2.f * v.z, 2.f * v.w);
    }
}

int main() {
    const double mbs[] = {5.5, 11, 22, 44, 88, 177};
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (double mb : mbs) {
        const long long n4 = (long long)(mb * 1024 * 1024 / 16);
        float4 *x, *y; hipMalloc(&x, n4 * 16); hipMalloc(&y, n4 * 16);
        hipMemset(x, 0, n4 * 16);
        long long blocks = (n4 + 255) / 256; const long long cap = n4 >= (8LL << 20) ? 16384 : 4096; if (blocks > cap) blocks = cap;
        for (int mode = 0; mode < 4; ++mode) {
            const int shift = mode & 1, reverse = mode >> 1;
            double tot = 0; const int reps = 20;
            for (int r = 0; r < reps + 3; ++r) {
                hipLaunchKernelGGL(pass_a, dim3(blocks), dim3(256), 0, 0, x, n4, out);
                hipEventRecord(e0, 0);
                hipLaunchKernelGGL(pass_b, dim3(blocks), dim3(256), 0, 0, x, y, n4, shift, reverse);
                hipEventRecord(e1, 0);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r >= 3) tot += ms;
            }
            printf("{\"map_MB\": %.1f, \"blocks\": %lld, \"shift\": %d, \"reverse\": %d, \"pass_b_us\": %.2f, \"GBps_on_2x_map\": %.0f}\n", mb, blocks, shift, reverse,
                   1e3 * tot / reps, 2 * mb * 1.048576e6 / (1e3 * tot / reps) / 1e3);
        }
        hipFree(x); hipFree(y);
    }
    return 0;
}
