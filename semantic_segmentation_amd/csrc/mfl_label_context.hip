// The reference's label-context block and the combine step behind it, fused (C ABI: mfl_label_context_forward):
//   a      = la_w * t + la_b                                   LinearActivation, AdvancedModel.py:52-56, 223
//   c      = ga_w * (Conv2d(L -> L*M, k x k, pad k/2)(a)) + ga_b   AdvancedModel.py:143-145, 232-233
//   smooth = min over the M mixtures, c viewed as [B][M][L][H][W]  BlockMinPooling, AdvancedModel.py:32-35, 235
//   logq   = LogSoftmax(log_softmax(unary) - smooth)               AdvancedModel.py:242-244
// The convolution is 81 * L * L * M multiply-adds per pixel (179 k at 21 labels, 5 mixtures): a dense implicit
// GEMM [L*M x L*k*k] . [L*k*k x pixels] that runs on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate).
//
// Layout of the GEMM
//   rows     r = 32 m + l  <->  output channel m*L + l (rows l >= L of a block are zero): the M mixtures of a
//            label sit in the same register of M accumulator blocks, so the min over mixtures is register-wise
//   K        kk = (c * k + ky) * k + kx, two per matrix instruction (lane half h takes 2j + h)
//   columns  32 consecutive pixels of an image row per tile, kTilesPerWave tiles per wave
//   A        weights, permuted to [kk][32 M] once per call (label_context_weights_kernel), staged through LDS in
//            chunks of kChunk K-values shared by the workgroup's waves
//   B        a, precomputed once per call into zero-padded planes (label_context_activation_kernel): the
//            convolution's zero padding applies to a (not to t), and a padded plane makes every load unguarded;
//            lane (p, h) reads apad[koff[kk] + pixel], koff from LDS
// Epilogue: register v of block m of lane (p, h) is output channel m*L + l, l = 8 (v/4) + 4 h + v%4, of the lane's
// pixel: bias, per-pixel affine, min over m, then the two softmaxes with the maxima and sums exchanged between
// the two lanes of a pixel.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfl.h"

#define MFL_CHECK_HIP(expr)                          \
    do {                                             \
        const hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return (int)e_;        \
    } while (0)

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

constexpr int kWavesPerGroup = 4;
constexpr int kTilesPerWave = 1;    // (two tiles per wave, 160 accumulators: one wave per SIMD, 13.9 instead of 11.3 ms)
constexpr int kChunk = 32;          // K-values of the weights per LDS stage (even)
constexpr int kAhead = 4;           // k-steps whose activation loads are in flight together

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// apad[b][c][y + p][x + p] = la_w[c][y][x] * t[b][c][y][x] + la_b[c][y][x]; the borders stay zero (memset by the caller)
__global__ void label_context_activation_kernel(const float *__restrict__ t, const float *__restrict__ la_w,
                                                const float *__restrict__ la_b, float *__restrict__ apad, int B, int L, int H,
                                                int W, int Hp, int Wp, int p)
{
    const size_t n = (size_t)B * L * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t r = i / W;
        const int y = (int)(r % H);
        const size_t bc = r / H;
        const int c = (int)(bc % L);
        float v = t[i];
        if (la_w) v = fmaf(la_w[((size_t)c * H + y) * W + x], v, la_b[((size_t)c * H + y) * W + x]);
        apad[(bc * Hp + y + p) * Wp + x + p] = v;
    }
}

// w2[kk][32 m + l] = conv_w[m*L + l][kk] (kk = (c*k + ky)*k + kx as stored), zero for l >= L and for the padded
// K-values kk >= KK; koff[kk] = offset of tap kk in the padded planes (0 for padded K-values)
__global__ void label_context_weights_kernel(const float *__restrict__ conv_w, float *__restrict__ w2, int *__restrict__ koff,
                                             int L, int M, int ks, int KK, int KKp, int plane, int Wp)
{
    const int R = 32 * M;
    const size_t n = (size_t)KKp * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i / R), r = (int)(i - (size_t)kk * R);
        const int m = r >> 5, l = r & 31;
        w2[i] = (l < L && kk < KK) ? conv_w[((size_t)(m * L + l)) * KK + kk] : 0.f;
        if (r == 0) {
            int off = 0;
            if (kk < KK) {
                const int c = kk / (ks * ks), rem = kk - c * ks * ks;
                off = c * plane + (rem / ks) * Wp + rem % ks;
            }
            koff[kk] = off;
        }
    }
}

template <int M>
__global__ void __launch_bounds__(64 * kWavesPerGroup)
label_context_kernel(const float *__restrict__ apad, const float *__restrict__ w2, const int *__restrict__ koff,
                     const float *__restrict__ conv_b, const float *__restrict__ ga_w, const float *__restrict__ ga_b,
                     const float *__restrict__ unary, float *__restrict__ smooth, float *__restrict__ logq, int B, int L, int H,
                     int W, int Wp, int plane, int KKp, int tiles_per_row)
{
    constexpr int R = 32 * M;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *wl = lds;                                           // [kChunk][R]
    int *ko = reinterpret_cast<int *>(lds + kChunk * R);       // [KKp]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int HW = H * W;
    for (int i = threadIdx.x; i < KKp; i += blockDim.x) ko[i] = koff[i];
    // this wave's tiles: 32 consecutive pixels of an image row each (the last tile of a row may run past
    // the row's end: those lanes compute on clamped pixels and store nothing)
    const long long ntiles = (long long)B * H * tiles_per_row;
    long long tile0 = ((long long)blockIdx.x * kWavesPerGroup + wave) * kTilesPerWave;
    int base[kTilesPerWave], pixl[kTilesPerWave], img[kTilesPerWave];
    bool ok[kTilesPerWave];
#pragma unroll
    for (int tt = 0; tt < kTilesPerWave; ++tt) {
        const long long tile = tile0 + tt < ntiles ? tile0 + tt : ntiles - 1;
        const int tx = (int)(tile % tiles_per_row);
        const long long ry = tile / tiles_per_row;
        const int y = (int)(ry % H), b = (int)(ry / H);
        const int x = tx * 32 + p;
        ok[tt] = tile0 + tt < ntiles && x < W;
        const int xc = x < W ? x : W - 1;
        img[tt] = b;
        pixl[tt] = y * W + xc;
        base[tt] = b * L * plane + y * Wp + xc;              // window origin of the pixel in the padded planes
    }
    f16v d[kTilesPerWave][M];
#pragma unroll
    for (int tt = 0; tt < kTilesPerWave; ++tt)
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) d[tt][m][v] = 0.f;

    for (int k0 = 0; k0 < KKp; k0 += kChunk) {
        __syncthreads();                                       // the previous chunk's readers are done (first pass: ko is complete)
        {
            const float4 *src = reinterpret_cast<const float4 *>(w2 + (size_t)k0 * R);
            float4 *dst = reinterpret_cast<float4 *>(wl);
            for (int i = threadIdx.x; i < kChunk * R / 4; i += blockDim.x) dst[i] = src[i];
        }
        __syncthreads();
        for (int j0 = 0; j0 < kChunk / 2; j0 += kAhead) {
            float bv[kAhead][kTilesPerWave];
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const int off = ko[k0 + 2 * (j0 + jj) + h];
#pragma unroll
                for (int tt = 0; tt < kTilesPerWave; ++tt) bv[jj][tt] = apad[base[tt] + off];
            }
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const float *row = wl + (2 * (j0 + jj) + h) * R + p;
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    const float a = row[32 * m];
#pragma unroll
                    for (int tt = 0; tt < kTilesPerWave; ++tt)
                        d[tt][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[jj][tt], d[tt][m], 0, 0, 0);
                }
            }
        }
    }

    auto other_half = [&](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
    };
#pragma unroll
    for (int tt = 0; tt < kTilesPerWave; ++tt) {
        const int l0 = pixl[tt];
        float sm[16];
        float mu = -INFINITY;
        float u[16];
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int l = 8 * (v / 4) + 4 * h + (v % 4);
            const bool valid = l < L;
            const int lc = valid ? l : L - 1;
            float mn = INFINITY;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int oc = m * L + lc;
                float c = d[tt][m][v] + (conv_b ? conv_b[oc] : 0.f);
                if (ga_w) c = fmaf(ga_w[(size_t)oc * HW + l0], c, ga_b[(size_t)oc * HW + l0]);
                mn = fminf(mn, c);
            }
            sm[v] = mn;
            u[v] = valid ? unary[((size_t)img[tt] * L + lc) * HW + l0] : -INFINITY;
            mu = fmaxf(mu, u[v]);
        }
        mu = fmaxf(mu, other_half(mu));
        float su = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) su += __expf(u[v] - mu);
        su += other_half(su);
        const float lse_u = mu + __logf(su);
        float mv = -INFINITY;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            u[v] = (u[v] - lse_u) - sm[v];        // -inf for the labels past L
            mv = fmaxf(mv, u[v]);
        }
        mv = fmaxf(mv, other_half(mv));
        float sv = 0.f;
#pragma unroll
        for (int v = 0; v < 16; ++v) sv += __expf(u[v] - mv);
        sv += other_half(sv);
        const float lse_v = mv + __logf(sv);
        if (ok[tt]) {
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int l = 8 * (v / 4) + 4 * h + (v % 4);
                if (l < L) {
                    const size_t o = ((size_t)img[tt] * L + l) * HW + l0;
                    if (smooth) smooth[o] = sm[v];
                    if (logq) logq[o] = u[v] - lse_v;
                }
            }
        }
    }
}

struct ContextWorkspace {
    size_t apad, w2, koff, total;
    int Hp, Wp, KK, KKp;
};

ContextWorkspace context_carve(int B, int L, int M, int H, int W, int ks)
{
    ContextWorkspace w{};
    w.Hp = H + ks - 1;
    w.Wp = W + ks - 1;
    w.KK = L * ks * ks;
    w.KKp = (w.KK + kChunk - 1) / kChunk * kChunk;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.apad = take(((size_t)B * L * w.Hp * w.Wp + 64) * 4);
    w.w2 = take((size_t)w.KKp * 32 * M * 4);
    w.koff = take((size_t)w.KKp * 4);
    w.total = off;
    return w;
}

bool context_supported(int L, int M, int ks) { return L >= 1 && L <= 32 && M >= 1 && M <= 5 && ks >= 1 && ks <= 15 && (ks & 1); }

template <int M>
int launch(const ContextWorkspace &ws, char *base, const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
           float *smooth, float *logq, int B, int L, int H, int W, hipStream_t st)
{
    const int tiles_per_row = (W + 31) / 32;
    const long long ntiles = (long long)B * H * tiles_per_row;
    const long long groups = (ntiles + kWavesPerGroup * kTilesPerWave - 1) / (kWavesPerGroup * kTilesPerWave);
    if (groups >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const size_t lds = (size_t)kChunk * 32 * M * 4 + (size_t)ws.KKp * 4;
    if (lds > 160 * 1024) return MFL_ERR_INVALID_ARGUMENT;
    // (the attribute belongs to the current device's copy of the kernel: set before every launch)
    MFL_CHECK_HIP(hipFuncSetAttribute((const void *)label_context_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(label_context_kernel<M>, dim3((unsigned)groups), dim3(64 * kWavesPerGroup), lds, st,
                       (const float *)(base + ws.apad), (const float *)(base + ws.w2), (const int *)(base + ws.koff), conv_b, ga_w, ga_b,
                       unary, smooth, logq, B, L, H, W, ws.Wp, ws.Hp * ws.Wp, ws.KKp, tiles_per_row);
    return (int)hipGetLastError();
}

}  // namespace

namespace mfl {

size_t label_context_workspace_bytes(int B, int L, int M, int H, int W, int ks)
{
    if (B <= 0 || H <= 0 || W <= 0 || !context_supported(L, M, ks)) return 0;
    if ((long long)B * L * (H + ks - 1) * (W + ks - 1) >= (1LL << 31)) return 0;
    return context_carve(B, L, M, H, W, ks).total;
}

}  // namespace mfl

extern "C" size_t mfl_label_context_workspace_bytes(int B, int L, int M, int H, int W, int ks)
{
    return mfl::label_context_workspace_bytes(B, L, M, H, W, ks);
}

extern "C" int mfl_label_context_forward(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                         const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                         float *smooth, float *logq, int B, int L, int M, int H, int W, int ks, void *workspace,
                                         size_t workspace_bytes, mfl_stream_t stream)
{
    if (!t || !conv_w || !unary || (!smooth && !logq)) return MFL_ERR_INVALID_ARGUMENT;
    if ((la_w == nullptr) != (la_b == nullptr) || (ga_w == nullptr) != (ga_b == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || H <= 0 || W <= 0 || !context_supported(L, M, ks)) return MFL_ERR_INVALID_ARGUMENT;
    if ((long long)B * L * (H + ks - 1) * (W + ks - 1) >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const ContextWorkspace ws = context_carve(B, L, M, H, W, ks);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    float *apad = (float *)(base + ws.apad);
    const size_t npad = (size_t)B * L * ws.Hp * ws.Wp + 64;
    MFL_CHECK_HIP(hipMemsetAsync(apad, 0, npad * 4, st));
    const size_t n = (size_t)B * L * H * W;
    const unsigned g1 = (unsigned)((n + 255) / 256 < 65535u * 8u ? (n + 255) / 256 : 65535u * 8u);
    hipLaunchKernelGGL(label_context_activation_kernel, dim3(g1), dim3(256), 0, st, t, la_w, la_b, apad, B, L, H, W, ws.Hp, ws.Wp,
                       ks / 2);
    const size_t nw = (size_t)ws.KKp * 32 * M;
    hipLaunchKernelGGL(label_context_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, conv_w,
                       (float *)(base + ws.w2), (int *)(base + ws.koff), L, M, ks, ws.KK, ws.KKp, ws.Hp * ws.Wp, ws.Wp);
    MFL_CHECK_HIP(hipGetLastError());
    switch (M) {
    case 1: return launch<1>(ws, base, conv_b, ga_w, ga_b, unary, smooth, logq, B, L, H, W, st);
    case 2: return launch<2>(ws, base, conv_b, ga_w, ga_b, unary, smooth, logq, B, L, H, W, st);
    case 3: return launch<3>(ws, base, conv_b, ga_w, ga_b, unary, smooth, logq, B, L, H, W, st);
    case 4: return launch<4>(ws, base, conv_b, ga_w, ga_b, unary, smooth, logq, B, L, H, W, st);
    case 5: return launch<5>(ws, base, conv_b, ga_w, ga_b, unary, smooth, logq, B, L, H, W, st);
    default: return MFL_ERR_INVALID_ARGUMENT;
    }
}
