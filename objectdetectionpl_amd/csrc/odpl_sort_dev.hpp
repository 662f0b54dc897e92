// Workgroup-wide bitonic sort of (key, index) pairs held in global (or LDS) memory.
#pragma once
#include "odpl_common.hpp"

namespace odpl {

// (key desc, idx asc): a strict total order, so the unstable network gives a unique result
__device__ __forceinline__ bool before(float ka, unsigned ia, float kb, unsigned ib) {
    return ka > kb || (ka == kb && ia < ib);
}

__device__ void bitonic_sort_global(float* key, unsigned* idx, int npad) {
    for (int k = 2; k <= npad; k <<= 1) {
        for (int lj = __ffs(k) - 2; lj >= 0; --lj) {            // j = 2^lj: shifts and masks instead of run-time divisions
            const int j = 1 << lj;
            for (int t = threadIdx.x; t < npad / 2; t += blockDim.x) {
                const int i = ((t >> lj) << (lj + 1)) + (t & (j - 1)), l = i + j;
                const bool up = ((i & k) == 0);
                const float ki = key[i], kl = key[l];
                const unsigned ii = idx[i], il = idx[l];
                const bool sw = up ? before(kl, il, ki, ii) : before(ki, ii, kl, il);
                if (sw) { key[i] = kl; key[l] = ki; idx[i] = il; idx[l] = ii; }
            }
            __syncthreads();
        }
    }
}


}  // namespace odpl
