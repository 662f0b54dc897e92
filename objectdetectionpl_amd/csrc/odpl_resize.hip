// Nearest-neighbour 2x upsampling of the YOLO necks on channels-last tensors, forward and backward -- SURVEY 8f rank 4 (the data
// movement between the heads' convolutions): model/YOLOV4.py's `Upsample` (utils/YoloV4Utils.py:191-206, F.interpolate(size=..., 'nearest')
// in training), nn.Upsample(scale_factor=2) of model/YOLOV3.py and utils/YoloV5Utils.py's neck.
//   forward : y[b, 2i+dy, 2j+dx, :] = x[b, i, j, :]                 one 16-B vector read, four written
//   backward: gx[b, i, j, :] = gy[b,2i,2j,:] + gy[b,2i,2j+1,:] + gy[b,2i+1,2j,:] + gy[b,2i+1,2j+1,:]   (fp32 sums, that order)
// ATen's channels-last kernels walk one element per thread with 64-bit index arithmetic (config 5, bf16: 233 us forward and 146 us
// backward per call for 13-52 MB maps); these are copy-shaped streams: lanes along the channels (a pixel's channels are one
// contiguous run), one input pixel row per blockIdx.y, 16-byte accesses.  The destination of the forward and the source of the
// backward take a pixel stride, so the result can be the channel slice of a concatenation (`torch.cat([lateral, up], 1)`).
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;

struct UpParams {
    const void* src; void* dst;
    int B, H, W;                 // the SMALL tensor's shape [B, C, H, W]
    int cv;                      // 16-byte vectors per pixel of the small tensor (C * elem / 16)
    long long big_px;            // pixel stride of the large tensor in 16-byte vectors (>= cv; == cv when dense)
    long long big_off;           // vector offset of the channel slice inside a large-tensor pixel
};

typedef float v4f __attribute__((ext_vector_type(4)));

// bf16 x 8 <-> fp32 x 8
__device__ __forceinline__ void unpack8(v4f v, float* f) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned u = __float_as_uint(v[k]); f[2 * k] = __uint_as_float(u << 16); f[2 * k + 1] = __uint_as_float(u & 0xffff0000u); }
}
__device__ __forceinline__ v4f pack8(const float* f) {
    v4f r;
#pragma unroll
    for (int k = 0; k < 4; ++k) r[k] = __uint_as_float(bf16_rne(f[2 * k]) | (bf16_rne(f[2 * k + 1]) << 16));
    return r;
}

__global__ __launch_bounds__(kT) void upsample2x_fwd_kernel(const UpParams P) {
    const int row = blockIdx.y;                                  // b * H + i
    const int b = row / P.H, i = row - b * P.H;
    const int nrow = P.W * P.cv;                                 // vectors of one input pixel row
    const v4f* src = reinterpret_cast<const v4f*>(P.src) + (size_t)row * nrow;
    v4f* d0 = reinterpret_cast<v4f*>(P.dst) + ((size_t)(b * 2 * P.H + 2 * i) * (2 * P.W)) * P.big_px + P.big_off;
    v4f* d1 = d0 + (size_t)(2 * P.W) * P.big_px;
    for (int t = blockIdx.x * kT + threadIdx.x; t < nrow; t += gridDim.x * kT) {
        const int j = t / P.cv, q = t - j * P.cv;
        const v4f v = __builtin_nontemporal_load(src + t);
        const size_t o = (size_t)(2 * j) * P.big_px + q;
        d0[o] = v; d0[o + P.big_px] = v; d1[o] = v; d1[o + P.big_px] = v;
    }
}

template <bool BF>
__global__ __launch_bounds__(kT) void upsample2x_bwd_kernel(const UpParams P) {
    const int row = blockIdx.y;
    const int b = row / P.H, i = row - b * P.H;
    const int nrow = P.W * P.cv;
    v4f* dst = reinterpret_cast<v4f*>(P.dst) + (size_t)row * nrow;
    const v4f* s0 = reinterpret_cast<const v4f*>(P.src) + ((size_t)(b * 2 * P.H + 2 * i) * (2 * P.W)) * P.big_px + P.big_off;
    const v4f* s1 = s0 + (size_t)(2 * P.W) * P.big_px;
    for (int t = blockIdx.x * kT + threadIdx.x; t < nrow; t += gridDim.x * kT) {
        const int j = t / P.cv, q = t - j * P.cv;
        const size_t o = (size_t)(2 * j) * P.big_px + q;
        const v4f a = __builtin_nontemporal_load(s0 + o), c = __builtin_nontemporal_load(s0 + o + P.big_px);
        const v4f d = __builtin_nontemporal_load(s1 + o), e = __builtin_nontemporal_load(s1 + o + P.big_px);
        if (!BF) {
            dst[t] = ((a + c) + d) + e;
        } else {
            float fa[8], fc[8], fd[8], fe[8], r[8];
            unpack8(a, fa); unpack8(c, fc); unpack8(d, fd); unpack8(e, fe);
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = ((fa[k] + fc[k]) + fd[k]) + fe[k];
            dst[t] = pack8(r);
        }
    }
}
}  // namespace

extern "C" int odpl_upsample2x_nearest(const void* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t bf16,
                                       int64_t big_pixel_stride, int64_t big_channel_offset, int32_t inverse, void* stream_) {
    const int per = bf16 ? 8 : 4;                                // elements per 16-byte vector
    if (!src || !dst || B < 1 || C < per || H < 1 || W < 1 || (C % per) || (long long)B * H > 65535) return ODPL_EINVAL;
    if (big_pixel_stride < C + big_channel_offset || big_channel_offset < 0 || (big_pixel_stride % per) || (big_channel_offset % per)) return ODPL_EINVAL;
    if ((long long)W * (C / per) > 0x7fffffffLL || (((uintptr_t)src | (uintptr_t)dst) & 15)) return ODPL_EINVAL;
    UpParams P;
    P.src = src; P.dst = dst; P.B = B; P.H = H; P.W = W; P.cv = C / per;
    P.big_px = big_pixel_stride / per; P.big_off = big_channel_offset / per;
    const int nrow = W * P.cv;
    int bx = (nrow + kT - 1) / kT; if (bx > 64) bx = 64;
    const dim3 grid(bx, B * H);
    hipStream_t stream = (hipStream_t)stream_;
    if (!inverse) hipLaunchKernelGGL(upsample2x_fwd_kernel, grid, dim3(kT), 0, stream, P);
    else if (bf16) hipLaunchKernelGGL(upsample2x_bwd_kernel<true>, grid, dim3(kT), 0, stream, P);
    else hipLaunchKernelGGL(upsample2x_bwd_kernel<false>, grid, dim3(kT), 0, stream, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
