// Device pieces of the BatchNorm + activation epilogues (odpl_bnact.hip).
#pragma once
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace odpl {

// Mish through ONE exponential: with n = e^z,  tanh(softplus(z)) = n(n+2) / (n(n+2) + 2)  and  sigmoid(z) = n / (1+n).
// (ATen evaluates tanh(log1p(exp(z))): three transcendentals, which makes the epilogue VALU-bound instead of
// HBM-bound.)  No cancellation: every term is positive; for z > 20 the ratio is 1 to fp32 precision.
// Divisions by the hardware reciprocal (v_rcp_f32, 1 ulp; both divisors are >= 1): HIP's `__fdividef` is an IEEE division
// here -- ten instructions (div_scale x2, rcp, four fma, div_fmas, div_fixup) -- and the backward kernels carry two per element,
// which is what kept them on the VALU issue limit instead of the HBM one.
__device__ __forceinline__ float div_rcp(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ __forceinline__ float mish_t(float z, float& n) {
    n = __expf(fminf(z, 20.0f));
    const float p = n * (n + 2.0f);
    return div_rcp(p, p + 2.0f);
}
// ACT is a template parameter of the streaming kernels: a run-time switch costs a scalar compare-and-branch per ELEMENT and cuts the
// loop body into basic blocks the scheduler cannot interleave.
template <int ACT>
__device__ __forceinline__ float act_fwd(float z) {
    constexpr int act = ACT;
    if (act == ODPL_ACT_MISH) { float n; return z * mish_t(z, n); }            // x * tanh(softplus(x))
    if (act == ODPL_ACT_LEAKY) return z > 0.f ? z : 0.1f * z;
    if (act == ODPL_ACT_RELU) return z > 0.f ? z : 0.f;
    return z;
}
template <int ACT>
__device__ __forceinline__ float act_grad(float z) {
    constexpr int act = ACT;
    if (act == ODPL_ACT_MISH) {                                                 // ts + z * sigmoid(z) * (1 - ts^2)  (ATen mish_backward)
        float n;
        const float ts = mish_t(z, n);
        const float sg = z > 20.0f ? 1.0f : div_rcp(n, 1.0f + n);
        return ts + z * sg * (1.0f - ts * ts);
    }
    if (act == ODPL_ACT_LEAKY) return z > 0.f ? 1.0f : 0.1f;
    if (act == ODPL_ACT_RELU) return z > 0.f ? 1.0f : 0.f;
    return 1.0f;
}

struct BnParams {
    const void* x; const void* gy; void* y; void* gx;  // activations / their gradients: fp32 or bf16 (template BF)
    const float* gamma; const float* beta;
    float* mean; float* invstd;                  // [C] batch statistics (saved for backward)
    float* running_mean; float* running_var;     // [C] or null
    const void* res;                             // optional residual added after the activation (same layout as y)
    double* part;                                // [nblk][C][2]
    float* dgamma; float* dbeta;                 // [C]
    long long* num_batches_tracked;              // nn.BatchNorm2d buffer, +1 per training forward (or null)
    long long R; int C, act, nblk, rows_per_blk, training;
    int res_pre;                                 // residual added BEFORE the activation (ResNet: relu(bn(x) + s)) instead of after it
    int slice_q;                                 // channel quads per workgroup of the reduce passes (grid.y slices of the row), or C/4
    long long y_rs4, gy_rs4;                     // row strides of y / grad_y in float4 units (== C/4 when dense)
    float eps, momentum;
    void* gu;                                    // bn_add_act backward: grad_y masked by the activation's derivative AT THE SAVED OUTPUT y (written by the reduce pass)
    unsigned char* flags;                        // bias_act_bwd: [R] bytes, set to 1 where the row of grad_x has a non-zero (pre-zeroed), or null
};

// 16 bytes (fp32) / 8 bytes (bf16) of a channels-last row at a byte offset from a workgroup-uniform base: the compiler keeps
// the base in scalar registers and the offset in ONE vector register (global_load ... v_off, s[base:base+1])
template <bool BF>
__device__ __forceinline__ float4 ld_quad_at(const char* base, unsigned byte_off) {
    if (!BF) return *reinterpret_cast<const float4*>(base + byte_off);
    const uint2 u = *reinterpret_cast<const uint2*>(base + byte_off);
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
}

}  // namespace odpl
