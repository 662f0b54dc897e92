// Device-side collate: the step BEFORE the hot path.  The reference's loader does, per image on a CPU
// worker, A.Resize(S, S) [cv2.INTER_LINEAR on uint8 HWC BGR] -> HorizontalFlip -> VerticalFlip -> float -> /255 -> CHW,
// and rewrites the COCO pixel boxes to normalised (img, cls, x, y, w, h) rows (dataset/Pascal.py:121-186).  Here the raw
// uint8 frames sit in HBM and one launch writes the whole [B, 3, S, S] fp32 batch (any dense strides), a second one the rows.
// Photometric augmentations that are per-channel point operations on the uint8 frame (RandomBrightnessContrast, RGBShift: both are
// cv2.LUT calls in albumentations) ride along as ONE composed 3 x 256-byte table per image, looked up after the resize.
// 8-bit bilinear = OpenCV's fixed-point algorithm (11-bit coefficients, int32 row pass, the >>4 / >>16 / +2 >> 2 vertical pass),
// bit-exact to oracle/collate.py; see that file for the pinning status (cv2 is absent: "parity unpinned" against cv2 itself).
// HBM-bound: reads h*w*3 source bytes (4 taps, L2-resident neighbours), writes S*S*3*4 bytes per image.
#include "odpl_common.hpp"
#include "../../include/odpl_hip.h"

namespace {
using namespace odpl;
constexpr int kT = 256;
constexpr int kMaxImg = 32;             // images per launch (kernel-argument block)

struct Img { const unsigned char* data; const unsigned char* lut; int h, w, flip, first, warp; double m[6]; };   // m: dst -> src affine map of the warp
struct CollateParams {
    Img img[kMaxImg];
    int n, S;
    float* x; long long xs[4];
};

__device__ __forceinline__ void coeff(int d, int dst, int src, int& s0, int& s1, int& a0, int& a1) {
    const double inv = (double)dst / (double)src, scale = 1.0 / inv;
    float f = (float)(((double)d + 0.5) * scale - 0.5);
    int s = (int)floorf(f);
    f -= (float)s;
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= src - 1) { f = 0.f; s = src - 1; }
    a1 = (int)rintf(f * 2048.0f);
    a0 = (int)rintf((1.0f - f) * 2048.0f);
    s0 = s; s1 = s + 1 < src ? s + 1 : src - 1;
}

// channel c of pixel (px, py) of the RESIZED and FLIPPED uint8 frame (A.Resize -> HorizontalFlip -> VerticalFlip)
__device__ __forceinline__ void resized_pixel(const Img& I, int S, int px, int py, int out[3]) {
    const int ry = (I.flip & 2) ? S - 1 - py : py, rx = (I.flip & 1) ? S - 1 - px : px;        // flips act on the resized frame
    int x0, x1, ax0, ax1, y0, y1, ay0, ay1;
    coeff(rx, S, I.w, x0, x1, ax0, ax1);
    coeff(ry, S, I.h, y0, y1, ay0, ay1);
    const unsigned char* r0 = I.data + (size_t)y0 * I.w * 3;
    const unsigned char* r1 = I.data + (size_t)y1 * I.w * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int h0 = r0[x0 * 3 + c] * ax0 + r0[x1 * 3 + c] * ax1;
        const int h1 = r1[x0 * 3 + c] * ax0 + r1[x1 * 3 + c] * ax1;
        const int v = (((ay0 * (h0 >> 4)) >> 16) + ((ay1 * (h1 >> 4)) >> 16) + 2) >> 2;
        out[c] = v < 0 ? 0 : (v > 255 ? 255 : v);
    }
}

// cv2.BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
__device__ __forceinline__ int reflect101(int p, int len) {
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

__global__ __launch_bounds__(kT) void collate_images_kernel(const CollateParams P) {
    const int S = P.S;
    const Img& I = P.img[blockIdx.y];
    const long long b = I.first;
    for (int t = blockIdx.x * kT + threadIdx.x; t < S * S; t += gridDim.x * kT) {
        const int oy = t / S, ox = t - oy * S;
        int v[3];
        if (!I.warp) {
            resized_pixel(I, S, ox, oy, v);
        } else {
            // A.ShiftScaleRotate = cv2.warpAffine(INTER_LINEAR, BORDER_REFLECT_101) on the resized frame: OpenCV's fixed-point
            // path -- source coordinates in 1/1024 (AB_BITS = 10) rounded to 1/32 pixel (INTER_BITS = 5), bilinear weights
            // (32 - fx)(32 - fy) * 32 of 2^15, result (sum + 2^14) >> 15
            const int adelta = (int)rint(I.m[0] * (double)ox * 1024.0), bdelta = (int)rint(I.m[3] * (double)ox * 1024.0);
            const int X0 = (int)rint((I.m[1] * (double)oy + I.m[2]) * 1024.0) + 16, Y0 = (int)rint((I.m[4] * (double)oy + I.m[5]) * 1024.0) + 16;
            const int X = (X0 + adelta) >> 5, Y = (Y0 + bdelta) >> 5;
            int sx = X >> 5, sy = Y >> 5;
            sx = sx < -32768 ? -32768 : (sx > 32767 ? 32767 : sx); sy = sy < -32768 ? -32768 : (sy > 32767 ? 32767 : sy);   // saturate_cast<short>
            const int fx = X & 31, fy = Y & 31;
            const int w00 = (32 - fx) * (32 - fy) * 32, w01 = fx * (32 - fy) * 32, w10 = (32 - fx) * fy * 32, w11 = fx * fy * 32;
            const int x0 = reflect101(sx, S), x1 = reflect101(sx + 1, S), y0 = reflect101(sy, S), y1 = reflect101(sy + 1, S);
            int p00[3], p01[3], p10[3], p11[3];
            resized_pixel(I, S, x0, y0, p00); resized_pixel(I, S, x1, y0, p01);
            resized_pixel(I, S, x0, y1, p10); resized_pixel(I, S, x1, y1, p11);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int r = (p00[c] * w00 + p01[c] * w01 + p10[c] * w10 + p11[c] * w11 + (1 << 14)) >> 15;
                v[c] = r < 0 ? 0 : (r > 255 ? 255 : r);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int u = v[c];
            if (I.lut) u = I.lut[c * 256 + u];               // photometric point operations, composed into one per-channel table by the host
            P.x[b * P.xs[0] + c * P.xs[1] + oy * P.xs[2] + ox * P.xs[3]] = (float)u / 255.0f;
        }
    }
}

struct TargetParams {
    const float* obj; const int* row_img; const int* dims;       // dims [B][3] = h, w, flip
    const double* affine;                                        // [B][7]: flag, then the FORWARD 2x3 matrix in resized-frame pixels; or null
    float* y; unsigned char* keep; int nT, S;
};

__global__ void collate_targets_kernel(const TargetParams P) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= P.nT) return;
    const int i = P.row_img[t];
    const double h = P.dims[i * 3], w = P.dims[i * 3 + 1]; const int fl = P.dims[i * 3 + 2];
    const float* o = P.obj + (size_t)t * 5;
    double x0 = (double)o[0] / w, y0 = (double)o[1] / h, x1 = ((double)o[0] + (double)o[2]) / w, y1 = ((double)o[1] + (double)o[3]) / h;
    if (fl & 1) { const double a = 1.0 - x1, b = 1.0 - x0; x0 = a; x1 = b; }
    if (fl & 2) { const double a = 1.0 - y1, b = 1.0 - y0; y0 = a; y1 = b; }
    const double S = P.S;
    bool keep = true;
    if (P.affine && P.affine[i * 7] != 0.0) {
        // albumentations 0.5.2 bbox_shift_scale_rotate: the four corners through the forward matrix (in pixels of the S x S
        // frame), the axis-aligned hull of the result; then Compose's filter_bboxes: clip to [0, 1], drop boxes left without area
        const double* M = P.affine + i * 7 + 1;
        const double cx[4] = {x0, x1, x1, x0}, cy[4] = {y0, y0, y1, y1};
        double nx0 = 0, nx1 = 0, ny0 = 0, ny1 = 0;
        for (int k = 0; k < 4; ++k) {
            const double px = cx[k] * S, py = cy[k] * S;
            const double tx = (M[0] * px + M[1] * py + M[2]) / S, ty = (M[3] * px + M[4] * py + M[5]) / S;
            if (k == 0) { nx0 = nx1 = tx; ny0 = ny1 = ty; }
            else { nx0 = fmin(nx0, tx); nx1 = fmax(nx1, tx); ny0 = fmin(ny0, ty); ny1 = fmax(ny1, ty); }
        }
        const double full = (nx1 - nx0) * (ny1 - ny0);
        x0 = fmin(fmax(nx0, 0.0), 1.0); x1 = fmin(fmax(nx1, 0.0), 1.0); y0 = fmin(fmax(ny0, 0.0), 1.0); y1 = fmin(fmax(ny1, 0.0), 1.0);
        const double clipped = (x1 - x0) * (y1 - y0);
        keep = full != 0.0 && clipped != 0.0 && clipped / full > 0.0;
    }
    float* y = P.y + (size_t)t * 6;
    y[0] = (float)i; y[1] = o[4];
    y[2] = (float)((x0 * S) / S); y[3] = (float)((y0 * S) / S);
    y[4] = (float)(((x1 - x0) * S) / S); y[5] = (float)(((y1 - y0) * S) / S);
    if (P.keep) P.keep[t] = keep ? 1 : 0;
}
}  // namespace

extern "C" int odpl_collate_images(const odpl_image_t* host_images, int32_t B, int32_t S, float* x,
                                   const int64_t* host_x_strides, void* stream_) {
    return odpl_collate_images_warp(host_images, nullptr, B, S, x, host_x_strides, stream_);
}

extern "C" int odpl_collate_images_warp(const odpl_image_t* host_images, const double* host_inverse_affine, int32_t B, int32_t S, float* x,
                                        const int64_t* host_x_strides, void* stream_) {
    if (!host_images || B < 0 || S < 1 || S > 8192 || !x || !host_x_strides) return ODPL_EINVAL;
    for (int i = 0; i < B; ++i)
        if (!host_images[i].data || host_images[i].h < 1 || host_images[i].w < 1 || host_images[i].h > 16384 || host_images[i].w > 16384) return ODPL_EINVAL;
    hipStream_t stream = (hipStream_t)stream_;
    int gx = (S * S + kT - 1) / kT; if (gx > 1024) gx = 1024;
    for (int i0 = 0; i0 < B; i0 += kMaxImg) {
        CollateParams P = {};
        P.n = B - i0 < kMaxImg ? B - i0 : kMaxImg; P.S = S; P.x = x;
        for (int k = 0; k < 4; ++k) P.xs[k] = host_x_strides[k];
        for (int k = 0; k < P.n; ++k) {
            const odpl_image_t& im = host_images[i0 + k];
            P.img[k].data = im.data; P.img[k].lut = im.lut; P.img[k].h = im.h; P.img[k].w = im.w; P.img[k].flip = im.flip; P.img[k].first = i0 + k;
            P.img[k].warp = (im.reserved & 1) && host_inverse_affine ? 1 : 0;
            for (int j = 0; j < 6; ++j) P.img[k].m[j] = P.img[k].warp ? host_inverse_affine[(size_t)(i0 + k) * 6 + j] : 0.0;
        }
        hipLaunchKernelGGL(collate_images_kernel, dim3(gx, P.n), dim3(kT), 0, stream, P);
        ODPL_HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int odpl_collate_targets(const float* objects, const int32_t* row_image, const int32_t* image_dims, int32_t nT,
                                    int32_t S, float* y, void* stream_) {
    return odpl_collate_targets_warp(objects, row_image, image_dims, nullptr, nT, S, y, nullptr, stream_);
}

extern "C" int odpl_collate_targets_warp(const float* objects, const int32_t* row_image, const int32_t* image_dims, const double* image_affine,
                                         int32_t nT, int32_t S, float* y, uint8_t* keep, void* stream_) {
    if (nT < 0 || S < 1) return ODPL_EINVAL;
    if (nT == 0) return 0;
    if (!objects || !row_image || !image_dims || !y) return ODPL_EINVAL;
    TargetParams P = {objects, row_image, image_dims, image_affine, y, keep, nT, S};
    hipLaunchKernelGGL(collate_targets_kernel, dim3((nT + 127) / 128), dim3(128), 0, (hipStream_t)stream_, P);
    ODPL_HIP_TRY(hipGetLastError());
    return 0;
}
