// Device-resident image preprocessing in front of the path (C ABI: mfl_preprocess_rgb8, mfl_resize_labels):
// what DPNModel._preprocess does per image on the host (AdvancedModel.py:487-498),
//   transforms.Resize((H, W))      Pillow's two-pass bilinear resampling of the 8-bit image
//   transforms.ToTensor()          HWC uint8 -> CHW float / 255
//   transforms.Normalize(mean, std)
// and the nearest-neighbour resize DPNModel.fit applies to the annotated images (AdvancedModel.py:265).
// Only the raw 8-bit image crosses to the device; the float tensor is produced there.
//
// The resampling is Pillow's (the library torchvision's Resize hands a PIL image to): per output position a window
// [xmin, xmin + n) of the input and fixed-point coefficients (22 fractional bits) computed by the host in double
// precision exactly as Pillow does, a horizontal pass whose result is ROUNDED TO 8 BITS, then a vertical pass, each
// out = clip((2^21 + sum coefficient * pixel) >> 22, 0, 255).  The integer arithmetic makes the 8-bit result
// identical to Pillow's; the float result then follows by the same three fp32 operations torchvision performs
// (x / 255, - mean, / std, each correctly rounded).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfl.h"

namespace {

constexpr int kPrecisionBits = 22;

__device__ __forceinline__ int clip8(int v)
{
    v >>= kPrecisionBits;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// tmp[y][xx][ch] = resampled along x; one thread per (y, xx), three channels.
__global__ void preprocess_horizontal_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ tmp,
                                             const int *__restrict__ coef, const int *__restrict__ bounds, int ksize, int H0,
                                             int W0, int W)
{
    const size_t n = (size_t)H0 * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), y = (int)(i / W);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const int *k = coef + (size_t)xx * ksize;
        const unsigned char *row = src + ((size_t)y * W0 + xmin) * 3;
        int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
        for (int x = 0; x < cnt; ++x) {
            const int kv = k[x];
            s0 += row[3 * x] * kv;
            s1 += row[3 * x + 1] * kv;
            s2 += row[3 * x + 2] * kv;
        }
        unsigned char *o = tmp + i * 3;
        o[0] = (unsigned char)clip8(s0);
        o[1] = (unsigned char)clip8(s1);
        o[2] = (unsigned char)clip8(s2);
    }
}

// out[ch][yy][x] = ((resampled along y) / 255 - mean[ch]) / std[ch]; one thread per (yy, x).  coef == NULL: no
// vertical resampling (the heights agree).
__global__ void preprocess_vertical_kernel(const unsigned char *__restrict__ tmp, float *__restrict__ out,
                                           const int *__restrict__ coef, const int *__restrict__ bounds, int ksize, int H, int W,
                                           float m0, float m1, float m2, float d0, float d1, float d2)
{
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), yy = (int)(i / W);
        int v0, v1, v2;
        if (coef) {
            const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
            const int *k = coef + (size_t)yy * ksize;
            int s0 = 1 << (kPrecisionBits - 1), s1 = s0, s2 = s0;
            for (int y = 0; y < cnt; ++y) {
                const unsigned char *p = tmp + ((size_t)(ymin + y) * W + x) * 3;
                const int kv = k[y];
                s0 += p[0] * kv;
                s1 += p[1] * kv;
                s2 += p[2] * kv;
            }
            v0 = clip8(s0); v1 = clip8(s1); v2 = clip8(s2);
        } else {
            const unsigned char *p = tmp + i * 3;
            v0 = p[0]; v1 = p[1]; v2 = p[2];
        }
        // (IEEE divisions: the three roundings torchvision's ToTensor and Normalize perform)
        out[i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v0, 255.f), m0), d0);
        out[n + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v1, 255.f), m1), d1);
        out[2 * n + i] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)v2, 255.f), m2), d2);
    }
}

// out[yy][xx] = src[yin[yy]][xin[xx]]
__global__ void resize_labels_kernel(const unsigned char *__restrict__ src, unsigned char *__restrict__ out,
                                     const int *__restrict__ yin, const int *__restrict__ xin, int W0, int H, int W)
{
    const size_t n = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        out[i] = src[(size_t)yin[i / W] * W0 + xin[i % W]];
}

unsigned grid_of(size_t n) { return (unsigned)((n + 255) / 256 < 65535u * 4u ? (n + 255) / 256 : 65535u * 4u); }

}  // namespace

extern "C" size_t mfl_preprocess_workspace_bytes(int H0, int W0, int H, int W)
{
    if (H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)H0 * W * 3 + 256;
}

extern "C" int mfl_preprocess_rgb8(const unsigned char *src, int H0, int W0, const int *xcoef, const int *xbounds, int xksize,
                                   const int *ycoef, const int *ybounds, int yksize, const float *mean, const float *stdev,
                                   float *out, int H, int W, void *workspace, size_t workspace_bytes, mfl_stream_t stream)
{
    if (!src || !out || !mean || !stdev || H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) return MFL_ERR_INVALID_ARGUMENT;
    // a pass is skipped exactly when its size does not change (as Pillow does); otherwise its tables are needed
    if ((W != W0) != (xcoef != nullptr) || (W != W0) != (xbounds != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((H != H0) != (ycoef != nullptr) || (H != H0) != (ybounds != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((xcoef && xksize <= 0) || (ycoef && yksize <= 0)) return MFL_ERR_INVALID_ARGUMENT;
    if (stdev[0] == 0.f || stdev[1] == 0.f || stdev[2] == 0.f) return MFL_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *mid = src;
    if (xcoef) {
        if (!workspace || workspace_bytes < mfl_preprocess_workspace_bytes(H0, W0, H, W)) return MFL_ERR_WORKSPACE;
        unsigned char *tmp = (unsigned char *)workspace;
        hipLaunchKernelGGL(preprocess_horizontal_kernel, dim3(grid_of((size_t)H0 * W)), dim3(256), 0, st, src, tmp, xcoef, xbounds, xksize, H0,
                           W0, W);
        mid = tmp;
    }
    hipLaunchKernelGGL(preprocess_vertical_kernel, dim3(grid_of((size_t)H * W)), dim3(256), 0, st, mid, out, ycoef, ybounds, yksize, H, W,
                       mean[0], mean[1], mean[2], stdev[0], stdev[1], stdev[2]);
    return (int)hipGetLastError();
}

extern "C" int mfl_resize_labels(const unsigned char *src, int H0, int W0, const int *yin, const int *xin, unsigned char *out, int H,
                                 int W, mfl_stream_t stream)
{
    if (!src || !yin || !xin || !out || H0 <= 0 || W0 <= 0 || H <= 0 || W <= 0) return MFL_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(resize_labels_kernel, dim3(grid_of((size_t)H * W)), dim3(256), 0, (hipStream_t)stream, src, out, yin, xin, W0, H, W);
    return (int)hipGetLastError();
}
