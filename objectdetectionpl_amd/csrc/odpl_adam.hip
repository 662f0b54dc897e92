// Adam in ONE launch over every parameter of the network (LightningFunc/optimizer.py:6-38 of the reference: torch.optim.Adam with
// `weight_decay` as an L2 term; the package's configure_optimizers returns this through optim.py).
//
// torch's fused Adam is a multi-tensor launch whose pointer lists must fit the 4 KB kernel-argument block: 64.4 M parameters in 362
// tensors take 10 launches of ~61 us = 0.61 ms per step, 1.8 GB of traffic at 2.9 TB/s (profiles/r03_bench_steady_state.csv).  Here the
// pointers that never change (parameter, exp_avg, exp_avg_sq, step) live in a DEVICE table built once, only the gradient pointers --
// autograd allocates new gradient tensors every step -- travel as kernel arguments (8 B each: 384 tensors per launch), and a block map
// turns the launch into one flat stream: workgroup -> (tensor, 4096-element chunk), 16-byte accesses, seven streams (read p, g, m, v;
// write p, m, v).  HBM bound: 7 x 4 B per parameter.
//
// The arithmetic is ATen's fused kernel (ATen/native/cuda/fused_adam_utils.cuh, ADAM_MODE::ORIGINAL, amsgrad = false), rounding for
// rounding: opmath float, the hyper-parameters double -- so the moment updates are evaluated in double and rounded to float once --
// and bias corrections from the device step count through the same pow / sqrt.  Which products that build contracted into fused
// multiply-adds was FITTED to its output by exact-rational emulation (tools/adam_dump.py + tools/adam_fit.py: exp_avg =
// fma(beta1, exp_avg, round((1 - beta1) * grad)) reproduces 60 000 elements x 3 steps with no mismatch, the other contraction differs
// in 0.2 % of them by one ulp; the update is (step_size * exp_avg) / denom with float divisions) and is written out explicitly here:
// this file, like the rest of the library, is built with -ffp-contract=off.  tests/test_gpu_optim.py holds it to torch bit for bit.
#include <math.h>
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;
constexpr int kBlockElems = ODPL_ADAM_BLOCK;        // elements per workgroup

struct AdamArgs {
    const uint64_t* table;                         // [n][4]: param, exp_avg, exp_avg_sq, step (device pointers)
    const unsigned* numel;                         // [n]
    const unsigned* blk0;                          // [n + 1]
    const unsigned* blk_tensor;                    // [n_blocks]
    double lr, beta1, beta2, eps, weight_decay;
    int t0, nt, maximize;                          // this launch covers tensors [t0, t0 + nt)
    unsigned b0;                                   // first block of tensor t0
    const float* grad[ODPL_ADAM_MAX_TENSORS];      // gradient of tensor t0 + i
};

__device__ __forceinline__ void adam_elem(float& param, float grad, float& exp_avg, float& exp_avg_sq, const AdamArgs& A,
                                          float bias_correction1, float bias_correction2_sqrt) {
    if (A.maximize) grad = -grad;
    if (A.weight_decay != 0) grad = (float)__builtin_fma((double)param, A.weight_decay, (double)grad);        // grad += param * weight_decay
    const double w1 = (1 - A.beta1) * (double)grad;                                                           // (rounded: not fused)
    exp_avg = (float)__builtin_fma(A.beta1, (double)exp_avg, w1);                                             // beta1 * exp_avg + (1 - beta1) * grad
    const double w2 = ((1 - A.beta2) * (double)grad) * (double)grad;
    exp_avg_sq = (float)__builtin_fma(A.beta2, (double)exp_avg_sq, w2);                                       // beta2 * exp_avg_sq + (1 - beta2) * grad * grad
    const float step_size = (float)(A.lr / (double)bias_correction1);
    const float denom = (float)((double)(sqrtf(exp_avg_sq) / bias_correction2_sqrt) + A.eps);
    param -= step_size * exp_avg / denom;
}

__global__ __launch_bounds__(kT) void adam_step_kernel(const AdamArgs A) {
    const unsigned blk = A.b0 + blockIdx.x;
    const unsigned t = A.blk_tensor[blk];
    const unsigned local = blk - A.blk0[t];
    const uint64_t* row = A.table + (size_t)t * 4;
    float* __restrict__ p = reinterpret_cast<float*>(row[0]);
    float* __restrict__ m = reinterpret_cast<float*>(row[1]);
    float* __restrict__ v = reinterpret_cast<float*>(row[2]);
    const float step_count = *reinterpret_cast<const float*>(row[3]);
    const float* __restrict__ g = A.grad[(int)t - A.t0];
    const unsigned n = A.numel[t];
    // std::pair<double, double> in ATen's functor, narrowed to opmath (float) at the call
    const double bc1 = 1 - pow(A.beta1, (double)step_count);
    const double bc2 = 1 - pow(A.beta2, (double)step_count);
    const float bias_correction1 = (float)bc1, bias_correction2_sqrt = (float)sqrt(bc2);
    const unsigned e0 = local * (unsigned)kBlockElems;
    const unsigned e1 = e0 + kBlockElems < n ? e0 + kBlockElems : n;
    const bool vec = ((((uintptr_t)p | (uintptr_t)m | (uintptr_t)v | (uintptr_t)g) & 15) == 0);
    if (vec) {
        const unsigned q0 = e0 >> 2, q1 = e1 >> 2;           // e0 is a multiple of 4096
        float4* p4 = reinterpret_cast<float4*>(p); float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
        const float4* g4 = reinterpret_cast<const float4*>(g);
        constexpr int U = kBlockElems / 4 / kT;             // float4 per thread: all loads of a thread are requested before the arithmetic
        float4 pp[U], gg[U], mm[U], vv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned q = q0 + threadIdx.x + u * kT, qc = q < q1 ? q : (q1 > q0 ? q1 - 1 : q0);
            if (q1 > q0) { pp[u] = p4[qc]; gg[u] = g4[qc]; mm[u] = m4[qc]; vv[u] = v4[qc]; }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned q = q0 + threadIdx.x + u * kT;
            if (q < q1) {
                adam_elem(pp[u].x, gg[u].x, mm[u].x, vv[u].x, A, bias_correction1, bias_correction2_sqrt);
                adam_elem(pp[u].y, gg[u].y, mm[u].y, vv[u].y, A, bias_correction1, bias_correction2_sqrt);
                adam_elem(pp[u].z, gg[u].z, mm[u].z, vv[u].z, A, bias_correction1, bias_correction2_sqrt);
                adam_elem(pp[u].w, gg[u].w, mm[u].w, vv[u].w, A, bias_correction1, bias_correction2_sqrt);
                p4[q] = pp[u]; m4[q] = mm[u]; v4[q] = vv[u];
            }
        }
        for (unsigned e = (q1 << 2) + threadIdx.x; e < e1; e += kT) {        // the last workgroup's <= 3 trailing elements
            float pe = p[e], me = m[e], ve = v[e];
            adam_elem(pe, g[e], me, ve, A, bias_correction1, bias_correction2_sqrt);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
    } else {
        for (unsigned e = e0 + threadIdx.x; e < e1; e += kT) {
            float pe = p[e], me = m[e], ve = v[e];
            adam_elem(pe, g[e], me, ve, A, bias_correction1, bias_correction2_sqrt);
            p[e] = pe; m[e] = me; v[e] = ve;
        }
    }
}
}  // namespace

extern "C" int odpl_adam_step(const odpl_adam_desc_t* d, const float* const* host_grads, void* stream_) {
    if (!d || !host_grads || d->n_tensors < 1 || d->n_blocks < 1 || !d->table || !d->numel || !d->blk0 || !d->blk_tensor || !d->host_blk0) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    AdamArgs A;
    A.table = d->table; A.numel = d->numel; A.blk0 = d->blk0; A.blk_tensor = d->blk_tensor;
    A.lr = d->lr; A.beta1 = d->beta1; A.beta2 = d->beta2; A.eps = d->eps; A.weight_decay = d->weight_decay; A.maximize = d->maximize;
    for (int t0 = 0; t0 < d->n_tensors; t0 += ODPL_ADAM_MAX_TENSORS) {
        const int nt = d->n_tensors - t0 < ODPL_ADAM_MAX_TENSORS ? d->n_tensors - t0 : ODPL_ADAM_MAX_TENSORS;
        A.t0 = t0; A.nt = nt; A.b0 = d->host_blk0[t0];
        const unsigned nblk = d->host_blk0[t0 + nt] - d->host_blk0[t0];
        for (int i = 0; i < nt; ++i) { if (!host_grads[t0 + i]) return ODPL_EINVAL; A.grad[i] = host_grads[t0 + i]; }
        for (int i = nt; i < ODPL_ADAM_MAX_TENSORS; ++i) A.grad[i] = nullptr;
        if (nblk == 0) continue;
        hipLaunchKernelGGL(adam_step_kernel, dim3(nblk), dim3(kT), 0, stream, A);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}
