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
//
// Backward (C ABI: mfl_label_context_backward; what autograd derives for phases 3-4, AdvancedModel.py:299): the
// forward runs again and keeps, per (image, label, pixel), the arg-min mixture and the convolution's value there
// (torch.min(dim) routes its gradient to that mixture only, AdvancedModel.py:35); then
//   pre          dz = G - exp(logq) * sum G, d unary, d smooth = -dz (+ a caller's gradient of smooth), the gradient of
//                the convolution's output (non-zero in the arg-min mixture only) into zero-padded planes, d ga_w, d ga_b
//   bias         d conv_b = plane sums of that gradient
//   data         d a = Conv2d(L*M -> L) of those planes with the flipped, transposed weights: an implicit GEMM with one
//                block of 32 rows, both operands staged through LDS plane by plane (context_dgrad_kernel)
//   affine       d t = la_w * d a, d la_w = sum_b d a * t, d la_b = sum_b d a
//   weights      d conv_w[o][i][ky][kx] = sum over pixels of (gradient plane o) * (padded activation plane i shifted by
//                the tap): a GEMM [L*M x pixels] . [pixels x L*k*k] on the same instruction, pixels split into slices
//                whose partial sums are added in a fixed order (context_wgrad_kernel, context_wgrad_reduce_kernel)
// Sums over images and pixels have a fixed order: the gradients are reproducible bit for bit.
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
// Tiles (32 pixels each) per wave of the fused forward: one (two tiles per wave, 160 accumulators, left one wave per
// SIMD: 13.9 instead of 11.3 ms).
constexpr int kTilesPerWave = 1;
constexpr int kChunk = 32;          // K-values of the weights per LDS stage (even)
constexpr int kMaxMixtures = 5;     // mixtures per label (the entry points reject more)
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
// packed: rows r = output channel m*L + l itself, R = 32 * ceil(L*M / 32) (fewer zero rows when L is not 32)
__global__ void label_context_weights_kernel(const float *__restrict__ conv_w, float *__restrict__ w2, int *__restrict__ koff,
                                             int L, int M, int ks, int KK, int KKp, int plane, int Wp, int R, int packed)
{
    const size_t n = (size_t)KKp * R;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(i / R), r = (int)(i - (size_t)kk * R);
        const int m = r >> 5, l = r & 31;
        if (packed) w2[i] = (r < L * M && kk < KK) ? conv_w[(size_t)r * KK + kk] : 0.f;
        else w2[i] = (l < L && kk < KK) ? conv_w[((size_t)(m * L + l)) * KK + kk] : 0.f;
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

// MODE 0: the fused forward epilogue (optionally keeping the arg-min mixture `amin` and the convolution's value
//         there `cmin`, for the backward)
// MODE 2: the fused forward with packed rows: M counts blocks of 32 rows, row r = output channel r; the mixtures of a
//         label then sit in different lanes and registers, so the accumulators pass through a per-wave LDS tile
//         before the min over the `mixtures` mixtures.  21 labels x 5 mixtures: 4 blocks instead of 5.
// Cin: channels of the padded input planes (L)
// (two waves per SIMD: the epilogue of one overlaps the matrix instructions of the other)
template <int M, int MODE>
__global__ void __launch_bounds__(64 * kWavesPerGroup, 2)
label_context_kernel(const float *__restrict__ apad, const float *__restrict__ w2, const int *__restrict__ koff,
                     const float *__restrict__ conv_b, const float *__restrict__ ga_w, const float *__restrict__ ga_b,
                     const float *__restrict__ unary, float *__restrict__ smooth, float *__restrict__ logq,
                     unsigned char *__restrict__ amin, float *__restrict__ cmin, int B, int L, int Cin, int H,
                     int W, int Wp, int plane, int KKp, int tiles_per_row, int mixtures)
{
    constexpr int R = 32 * M;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // two stages of weights [2][kChunk][R], then the tap offsets [KKp]
    int *ko = reinterpret_cast<int *>(lds + 2 * kChunk * R);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int HW = H * W;
    for (int i = threadIdx.x; i < KKp; i += blockDim.x) ko[i] = koff[i];
    // this wave's tiles: 32 consecutive pixels of an image row each (the last tile of a row may run past
    // the row's end: those lanes compute on clamped pixels and store nothing)
    const long long ntiles = (long long)B * H * tiles_per_row;
    // Tile order: the images of one tile position are consecutive, and each XCD (workgroups are dealt to the 8 XCDs
    // round robin) takes a contiguous range of positions: the affine's planes, shared by the images, and the rows that
    // vertically adjacent tiles share are then read through one XCD's L2.
    const long long wg = (long long)(blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8;
    long long tile0 = (wg * kWavesPerGroup + wave) * kTilesPerWave;
    int base[kTilesPerWave], pixl[kTilesPerWave], img[kTilesPerWave];
    bool ok[kTilesPerWave];
#pragma unroll
    for (int tt = 0; tt < kTilesPerWave; ++tt) {
        const long long tile = tile0 + tt < ntiles ? tile0 + tt : ntiles - 1;
        const int b = (int)(tile % B);
        const long long pos = tile / B;
        const int tx = (int)(pos % tiles_per_row), y = (int)(pos / tiles_per_row);
        const int x = tx * 32 + p;
        ok[tt] = tile0 + tt < ntiles && x < W;
        const int xc = x < W ? x : W - 1;
        img[tt] = b;
        pixl[tt] = y * W + xc;
        base[tt] = b * Cin * plane + y * Wp + xc;            // window origin of the pixel in the padded planes
    }
    // The accumulators start at the convolution's bias (register v of block m of lane (p, h) is row 32 m + 8 (v/4) + 4 h +
    // v%4: output channel m*L + l, or the row itself with packed rows), so the epilogue has no bias to fetch.
    f16v d[kTilesPerWave][M];
#pragma unroll
    for (int m = 0; m < M; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int l = 8 * (v / 4) + 4 * h + (v % 4);
            const int oc = MODE == 2 ? 32 * m + l : m * L + l;
            const bool row = MODE == 2 ? oc < L * mixtures : l < L;
            const float bias = (conv_b && row) ? conv_b[oc] : 0.f;
#pragma unroll
            for (int tt = 0; tt < kTilesPerWave; ++tt) d[tt][m][v] = bias;
        }

    // Software pipeline: the weights of chunk c + 1 travel global -> registers while chunk c is multiplied and reach
    // the other LDS stage after it (one barrier per chunk); the activations of group g + 1 (kAhead k-steps) are
    // loaded before the matrix instructions of group g.
    constexpr int kGroups = kChunk / (2 * kAhead);             // per chunk (even: the two activation buffers alternate)
    static_assert(kGroups % 2 == 0, "activation double buffer");
    constexpr int WPT = kChunk * R / 4 / (64 * kWavesPerGroup); // float4 pieces of a weight stage per thread
    static_assert(kChunk * R / 4 % (64 * kWavesPerGroup) == 0, "weight stage");
    const int nchunks = KKp / kChunk;
    float4 wr[WPT];
    {
        const float4 *src = reinterpret_cast<const float4 *>(w2);
        float4 *dst = reinterpret_cast<float4 *>(lds);
#pragma unroll
        for (int i = 0; i < WPT; ++i) dst[threadIdx.x + i * 64 * kWavesPerGroup] = src[threadIdx.x + i * 64 * kWavesPerGroup];
    }
    __syncthreads();                                           // stage 0 and the tap offsets are complete
    float bv[2][kAhead][kTilesPerWave];
#pragma unroll
    for (int jj = 0; jj < kAhead; ++jj) {
        const int off = ko[2 * jj + h];
#pragma unroll
        for (int tt = 0; tt < kTilesPerWave; ++tt) bv[0][jj][tt] = apad[base[tt] + off];
    }
    for (int c = 0; c < nchunks; ++c) {
        // (the last pass fetches and stages its own chunk again rather than branch: nothing reads that stage)
        const int cn = c + 1 < nchunks ? c + 1 : c;
        {
            const float4 *src = reinterpret_cast<const float4 *>(w2 + (size_t)cn * kChunk * R);
#pragma unroll
            for (int i = 0; i < WPT; ++i) wr[i] = src[threadIdx.x + i * 64 * kWavesPerGroup];
        }
        const float *wl = lds + (c & 1) * kChunk * R;
#pragma unroll
        for (int gg = 0; gg < kGroups; ++gg) {
            int knext = c * kChunk + (gg + 1) * 2 * kAhead;
            if (knext >= KKp) knext = 0;                       // (past the end: any valid group, unused)
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const int off = ko[knext + 2 * jj + h];
#pragma unroll
                for (int tt = 0; tt < kTilesPerWave; ++tt) {
#ifdef MFL_EXP_NOB
                    bv[(gg + 1) & 1][jj][tt] = __builtin_bit_cast(float, off + base[tt]);
#else
                    bv[(gg + 1) & 1][jj][tt] = apad[base[tt] + off];
#endif
                }
            }
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const float *row = wl + (2 * (gg * kAhead + jj) + h) * R + p;
#pragma unroll
                for (int m = 0; m < M; ++m) {
#ifdef MFL_EXP_NOA
                    const float a = __builtin_bit_cast(float, (int)(row - lds) + m);
#else
                    const float a = row[32 * m];
#endif
#pragma unroll
                    for (int tt = 0; tt < kTilesPerWave; ++tt)
                        d[tt][m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv[gg & 1][jj][tt], d[tt][m], 0, 0, 0);
                }
            }
        }
        {
            float4 *dst = reinterpret_cast<float4 *>(lds + ((c + 1) & 1) * kChunk * R);
#pragma unroll
            for (int i = 0; i < WPT; ++i) dst[threadIdx.x + i * 64 * kWavesPerGroup] = wr[i];
        }
        __syncthreads();                                       // stage (c + 1) & 1 is complete; stage c & 1 may be overwritten
    }
#ifdef MFL_EXP_NOEPI
    if (MODE != 1) {
        float acc = 0.f;
#pragma unroll
        for (int m = 0; m < M; ++m)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc += d[0][m][v];
        float *dst = logq ? logq : smooth;
        if (ok[0] && h == 0 && dst) dst[(size_t)img[0] * L * HW + pixl[0]] = acc;
        return;
    }
#endif

    auto other_half = [&](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
    };
#pragma unroll
    for (int tt = 0; tt < kTilesPerWave; ++tt) {
        const int l0 = pixl[tt];
        float sm[16];
        float mu = -INFINITY;
        float u[16];
        if constexpr (MODE == 2) {
            // rows r = output channel: all accumulators pass through this wave's LDS tile [32 M rows][32 pixels] (it reuses
            // the weight stages: every wave has passed the loop's last barrier), and a lane collects the mixtures of its
            // labels from there.  The loop over the mixtures is unrolled to the maximum with the index clamped (a repeated
            // mixture never wins the strict comparison), so that the loads of the affine's planes are independent of any
            // branch and in flight together.
            float *tile = lds + wave * (R * 32);
            float cvat[16];
            int mat[16];
#pragma unroll
            for (int nb = 0; nb < M; ++nb)
#pragma unroll
                for (int v = 0; v < 16; ++v) tile[(32 * nb + 8 * (v / 4) + 4 * h + (v % 4)) * 32 + p] = d[tt][nb][v];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int l = 8 * (v / 4) + 4 * h + (v % 4);
                const int lc = l < L ? l : L - 1;
#pragma unroll
                for (int m = 0; m < kMaxMixtures; ++m) {
                    const int mc = m < mixtures ? m : mixtures - 1;
                    const int oc = mc * L + lc;
                    const float cv = tile[oc * 32 + p];
                    float c = cv;
                    if (ga_w) c = fmaf(ga_w[(size_t)oc * HW + l0], c, ga_b[(size_t)oc * HW + l0]);
                    // (mixtures arrive in increasing m: the first minimum wins, as torch.min(dim))
                    if (m == 0 || c < sm[v]) { sm[v] = c; mat[v] = mc; cvat[v] = cv; }
                }
            }
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int l = 8 * (v / 4) + 4 * h + (v % 4);
                const bool valid = l < L;
                const int lc = valid ? l : L - 1;
                if (amin && valid && ok[tt]) {
                    const size_t o = ((size_t)img[tt] * L + lc) * HW + l0;
                    amin[o] = (unsigned char)mat[v];
                    cmin[o] = cvat[v];
                }
                u[v] = valid ? unary[((size_t)img[tt] * L + lc) * HW + l0] : -INFINITY;
                mu = fmaxf(mu, u[v]);
            }
        } else {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int l = 8 * (v / 4) + 4 * h + (v % 4);
            const bool valid = l < L;
            const int lc = valid ? l : L - 1;
            float mn = INFINITY, cv_at = 0.f;
            int m_at = 0;
#pragma unroll
            for (int m = 0; m < M; ++m) {
                const int oc = m * L + lc;
                const float cv = d[tt][m][v];
                float c = cv;
                if (ga_w) c = fmaf(ga_w[(size_t)oc * HW + l0], c, ga_b[(size_t)oc * HW + l0]);
                if (m == 0 || c < mn) { mn = c; m_at = m; cv_at = cv; }      // (the first minimum, as torch.min(dim))
            }
            sm[v] = mn;
            if (amin && valid && ok[tt]) {
                const size_t o = ((size_t)img[tt] * L + lc) * HW + l0;
                amin[o] = (unsigned char)m_at;
                cmin[o] = cv_at;
            }
            u[v] = valid ? unary[((size_t)img[tt] * L + lc) * HW + l0] : -INFINITY;
            mu = fmaxf(mu, u[v]);
        }
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

// ------------------------------------------------------------------ backward kernels

// One thread per pixel of the image plane, looping over the batch (so the sums over images have a fixed order).
//   dz = G - exp(logq) * sum_c G (LogSoftmax, AdvancedModel.py:244); d unary = dz - softmax(unary) * sum_c dz (:242);
//   d smooth = gs - dz; the arg-min mixture m* of (b, l, pixel) receives it (BlockMinPooling, :35):
//   gpad[b][m* L + l][y + p][x + p] = ga_w * d smooth (the planes were zeroed: the other mixtures stay 0);
//   d ga_w[m* L + l][pixel] += d smooth * conv value there, d ga_b += d smooth (zeroed by the caller).
// grad_logq / grad_smooth: either may be NULL (not both).
// LP: the label count rounded up to a multiple of 8 (one build each): a thread keeps the pixel's gradient, unary and
// log-marginal of every label in registers, so each is loaded once and all loads of an image are in flight together
// (a first version walked the labels three times with loads inside each loop: 1.8 ms at 21 labels, 512 x 512, batch 8).
template <int LP>
__global__ void __launch_bounds__(256)
context_pre_kernel(const float *__restrict__ grad_logq, const float *__restrict__ grad_smooth, const float *__restrict__ logq,
                   const float *__restrict__ unary, const unsigned char *__restrict__ amin, const float *__restrict__ cmin,
                   const float *__restrict__ ga_w, float *__restrict__ gpad, float *__restrict__ d_unary, float *__restrict__ d_ga_w,
                   float *__restrict__ d_ga_b, int B, int L, int M, int H, int W, int HpG, int WpG, int p)
{
    const int HW = H * W;
    const int pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= HW) return;
    const int y = pix / W, x = pix - y * W;
    const size_t gplane = (size_t)HpG * WpG;
    const size_t gpix = (size_t)(y + p) * WpG + x + p;
    for (int b = 0; b < B; ++b) {
        const size_t base = (size_t)b * L * HW + pix;
        float g[LP], u[LP], e[LP], gs[LP], cm[LP];
        int am[LP];
#pragma unroll
        for (int l = 0; l < LP; ++l) {
            const size_t o = base + (size_t)(l < L ? l : L - 1) * HW;
            g[l] = (grad_logq && l < L) ? grad_logq[o] : 0.f;
            u[l] = l < L ? unary[o] : -INFINITY;
            e[l] = (grad_logq && l < L) ? logq[o] : -INFINITY;
            gs[l] = (grad_smooth && l < L) ? grad_smooth[o] : 0.f;
            am[l] = amin[o];
            cm[l] = cmin[o];
        }
        float sg = 0.f, mx = -INFINITY;
#pragma unroll
        for (int l = 0; l < LP; ++l) { sg += g[l]; mx = fmaxf(mx, u[l]); }
        float sdz = 0.f, se = 0.f;
#pragma unroll
        for (int l = 0; l < LP; ++l) {
            g[l] = g[l] - __expf(e[l]) * sg;           // dz (0 past L: exp(-inf) = 0)
            u[l] = __expf(u[l] - mx);
            sdz += g[l];
            se += u[l];
        }
        const float rse = 1.f / se;
#pragma unroll
        for (int l = 0; l < LP; ++l) {
            if (l < L) {
                const size_t o = base + (size_t)l * HW;
                d_unary[o] = g[l] - u[l] * rse * sdz;
                const float ds = gs[l] - g[l];
                const int oc = am[l] * L + l;
                const size_t oa = (size_t)oc * HW + pix;
                float dconv = ds;
                if (ga_w) {
                    dconv = ga_w[oa] * ds;
                    d_ga_w[oa] = fmaf(ds, cm[l], d_ga_w[oa]);
                    d_ga_b[oa] += ds;
                }
                gpad[((size_t)b * L * M + oc) * gplane + gpix] = dconv;
            }
        }
    }
}

// d conv_b[oc] = sum over images and pixels of the gradient planes (the padding adds zeros): one workgroup per
// (output channel, image) sums a plane (a fixed strided order per thread, then a fixed tree) into part[b][oc];
// the second kernel adds the images in order.
__global__ void __launch_bounds__(256)
context_bias_grad_kernel(const float *__restrict__ gpad, float *__restrict__ part, int LM, size_t gplane)
{
    __shared__ float sh[256];
    const int oc = blockIdx.x, b = blockIdx.y;
    const float *pl = gpad + ((size_t)b * LM + oc) * gplane;
    float s = 0.f;
    for (size_t i = threadIdx.x; i < gplane; i += 256) s += pl[i];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[(size_t)b * LM + oc] = sh[0];
}

__global__ void context_bias_grad_sum_kernel(const float *__restrict__ part, float *__restrict__ d_conv_b, int B, int LM)
{
    const int oc = blockIdx.x * blockDim.x + threadIdx.x;
    if (oc >= LM) return;
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += part[(size_t)b * LM + oc];
    d_conv_b[oc] = s;
}

// Weights of the data gradient: d a = Conv2d(L*M -> L, ks x ks, pad ks/2) of the gradient planes with
// wt[kk][i] = conv_w[o][i][ks-1-ky][ks-1-kx], kk = (o*ks + ky)*ks + kx (zero for i >= L and kk >= KK2)
__global__ void context_dgrad_weights_kernel(const float *__restrict__ conv_w, float *__restrict__ wt, int L, int LM, int ks, int KK2,
                                             int KK2p)
{
    const size_t n = (size_t)KK2p * 32;
    const int kk2 = ks * ks;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += (size_t)gridDim.x * blockDim.x) {
        const int kk = (int)(idx >> 5), i = (int)(idx & 31);
        float v = 0.f;
        if (kk < KK2 && i < L) {
            const int o = kk / kk2, rem = kk - o * kk2, ky = rem / ks, kx = rem - ky * ks;
            v = conv_w[(((size_t)o * L + i) * ks + (ks - 1 - ky)) * ks + (ks - 1 - kx)];
        }
        wt[idx] = v;
    }
}

// Data gradient: d a[i][y][x] = sum over (o, ky, kx) of wt[(o, ky, kx)][i] * gpad[o][y + ky][x + kx], a convolution
// L*M -> L channels whose GEMM has ONE block of 32 rows (the L labels), so an operand fetched from memory would feed a
// single matrix instruction.  A workgroup of four waves therefore takes a strip of 512 pixels of an image row (four
// 32-pixel tiles per wave) and walks the L*M gradient planes: per plane the k rows x (512 + k - 1) columns the strip's
// windows reach and the plane's k*k x 32 weights go through LDS (coalesced loads, the next plane's in registers while this
// one is multiplied, two LDS stages, one barrier per plane), and every matrix instruction reads both operands from there:
// lane (p, h) supplies row / pixel p at tap 2j + h.  An odd tap count gets one more tap of zero weights.
// (The first version ran the forward's kernel with the pixels' values loaded from global memory, one load per matrix
// instruction: 11.1 ms at 21 labels, 5 mixtures, 9 x 9, 512 x 512, batch 8.)
constexpr int kDgTilesPerWave = 4;
constexpr int kDgStrip = kWavesPerGroup * kDgTilesPerWave * 32;      // pixels of a strip

template <int KS>
struct DgCfg {
    static constexpr int PW = kDgStrip + KS - 1;                     // columns of a plane's patch
    static constexpr int NP = KS * PW;                               // ... its elements
    static constexpr int KT = KS * KS, KTP = (KT + 1) / 2 * 2;       // taps, rounded up to even
    static constexpr int NW = KTP * 32;                              // weights of a plane
    static constexpr int PPT = (NP + 64 * kWavesPerGroup - 1) / (64 * kWavesPerGroup);
    static constexpr int WPT = (NW + 64 * kWavesPerGroup - 1) / (64 * kWavesPerGroup);
    static constexpr size_t lds_bytes = (size_t)(2 * NP + 2 * NW + KTP) * 4;
};

template <int KS>
__global__ void __launch_bounds__(64 * kWavesPerGroup, 2)
context_dgrad_kernel(const float *__restrict__ gpad, const float *__restrict__ wt, float *__restrict__ out, int B, int L, int LM,
                     int H, int W, int WpG, size_t gplane, int strips_per_row)
{
    using Cfg = DgCfg<KS>;
    constexpr int NT = 64 * kWavesPerGroup;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *patch = lds;                                  // [2][NP]
    float *wl = lds + 2 * Cfg::NP;                       // [2][NW]
    int *toff = reinterpret_cast<int *>(lds + 2 * Cfg::NP + 2 * Cfg::NW);   // [KTP] offset of a tap in the patch
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = lane & 31, h = lane >> 5;
    const int strip = blockIdx.x % strips_per_row;
    const long long ry = blockIdx.x / strips_per_row;
    const int y = (int)(ry % H), b = (int)(ry / H);
    const int x0 = strip * kDgStrip;
    for (int kk = tid; kk < Cfg::KTP; kk += NT) toff[kk] = kk < Cfg::KT ? (kk / KS) * Cfg::PW + kk % KS : 0;
    float rp[Cfg::PPT], rw[Cfg::WPT];
    const float *gb = gpad + (size_t)b * LM * gplane + (size_t)y * WpG + x0;
    auto fetch = [&](int o) {
        const float *src = gb + (size_t)o * gplane;
#pragma unroll
        for (int i = 0; i < Cfg::PPT; ++i) {
            const int idx = tid + i * NT;
            const int r = idx / Cfg::PW, c = idx - r * Cfg::PW;
            // (columns past the plane's pitch belong to pixels past the image row: zero)
            rp[i] = (idx < Cfg::NP && x0 + c < WpG) ? src[(size_t)r * WpG + c] : 0.f;
        }
        const float *wsrc = wt + (size_t)o * Cfg::KT * 32;
#pragma unroll
        for (int i = 0; i < Cfg::WPT; ++i) {
            const int idx = tid + i * NT;
            rw[i] = idx < Cfg::KT * 32 ? wsrc[idx] : 0.f;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < Cfg::PPT; ++i) {
            const int idx = tid + i * NT;
            if (idx < Cfg::NP) patch[buf * Cfg::NP + idx] = rp[i];
        }
#pragma unroll
        for (int i = 0; i < Cfg::WPT; ++i) {
            const int idx = tid + i * NT;
            if (idx < Cfg::NW) wl[buf * Cfg::NW + idx] = rw[i];
        }
    };
    f16v d[kDgTilesPerWave];
#pragma unroll
    for (int tt = 0; tt < kDgTilesPerWave; ++tt)
#pragma unroll
        for (int v = 0; v < 16; ++v) d[tt][v] = 0.f;
    fetch(0);
    stage(0);
    __syncthreads();
    const int px = wave * kDgTilesPerWave * 32 + p;      // the lane's pixel of tile 0 within the strip
    for (int o = 0; o < LM; ++o) {
        // (the last pass fetches and stages its own plane again rather than branch: nothing reads that stage)
        fetch(o + 1 < LM ? o + 1 : o);
        const float *pb = patch + (o & 1) * Cfg::NP + px;
        const float *wb = wl + (o & 1) * Cfg::NW + p;
#pragma unroll 2
        for (int j = 0; j < Cfg::KTP / 2; ++j) {
            const int kk = 2 * j + h;
            const float a = wb[kk * 32];
            const float *row = pb + toff[kk];
#pragma unroll
            for (int tt = 0; tt < kDgTilesPerWave; ++tt) d[tt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, row[tt * 32], d[tt], 0, 0, 0);
        }
        stage((o + 1) & 1);
        __syncthreads();
    }
    // register v of lane (p, h): label 8 (v/4) + 4 h + v%4 of the lane's pixel
#pragma unroll
    for (int tt = 0; tt < kDgTilesPerWave; ++tt) {
        const int x = x0 + px + tt * 32;
        if (x >= W) continue;
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int l = 8 * (v / 4) + 4 * h + (v % 4);
            if (l < L) out[(((size_t)b * L + l) * H + y) * W + x] = d[tt][v];
        }
    }
}

template <int KS>
int launch_dgrad(const float *gpad, const float *wt, float *out, int B, int L, int LM, int H, int W, int WpG, size_t gplane, hipStream_t st)
{
    const int strips = (W + kDgStrip - 1) / kDgStrip;
    const long long blocks = (long long)B * H * strips;
    if (blocks >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    MFL_CHECK_HIP(hipFuncSetAttribute((const void *)context_dgrad_kernel<KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(context_dgrad_kernel<KS>, dim3((unsigned)blocks), dim3(64 * kWavesPerGroup), DgCfg<KS>::lds_bytes, st, gpad, wt, out, B,
                       L, LM, H, W, WpG, gplane, strips);
    return (int)hipGetLastError();
}

// d t = la_w * d a;  d la_w = sum_b d a * t;  d la_b = sum_b d a  (LinearActivation, AdvancedModel.py:52-56).  One thread per
// (label, pixel), looping over the batch.  la_w NULL: d t = d a only.
__global__ void __launch_bounds__(256)
context_affine_grad_kernel(const float *__restrict__ da, const float *__restrict__ t, const float *__restrict__ la_w,
                           float *__restrict__ d_t, float *__restrict__ d_la_w, float *__restrict__ d_la_b, int B, size_t LHW)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= LHW) return;
    const float w = la_w ? la_w[i] : 1.f;
    float sw = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = da[(size_t)b * LHW + i];
        d_t[(size_t)b * LHW + i] = w * g;
        sw = fmaf(g, t[(size_t)b * LHW + i], sw);
        sb += g;
    }
    if (la_w) { d_la_w[i] = sw; d_la_b[i] = sb; }
}

// Weight gradient: D[row][col] = sum over pixels of g[row][pixel] * a[col][pixel], rows r = output channel (gradient plane
// r; NB blocks of 32 rows), columns = taps kk = (i*ks + ky)*ks + kx: a[col][pixel] is the padded activation plane i
// shifted by (ky, kx).  A workgroup of four waves takes four consecutive column blocks (one per wave) over a slice of
// row tiles (64 consecutive pixels of one image row) and stages every tile through LDS with coalesced loads:
//   As[row][64]       the tile of all L*M gradient planes (every wave multiplies the same rows)
//   Bs[(i, ky)][64 + ks - 1]   the activation rows the workgroup's 128 columns read; column (i, ky, kx) is row (i, ky)
//                     shifted by kx
// and prefetches the next tile into registers while the matrix instructions of the current one run.  Per matrix
// instruction lane (p, h) supplies row / column p at pixel 32 h + j of the tile.  The gradient planes have a row pitch
// that covers the last, partial tile of an image row with zeros; what the activation planes hold there is multiplied
// by those zeros.  (A first version let every lane stream its own plane row from global memory: 25 ms at 21 labels,
// 5 mixtures, 512 x 512, batch 8, bound by the 64 cache lines each such load touches.)
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f4a __attribute__((ext_vector_type(4)));

constexpr int kWgPitchA = 68;            // floats per As row: 16-byte aligned, lanes p = 0 .. 7 of a 16-byte read cover all banks

template <int NB>
__global__ void __launch_bounds__(64 * kWavesPerGroup)
context_wgrad_kernel(const float *__restrict__ gpad, const float *__restrict__ apad, float *__restrict__ part, int B, int L, int LM,
                     int H, int W, int ks, int Wp, int plane, int WpG, int gplane, int KK, int ncb, int ncbg, int nsl,
                     int tiles_per_slice, int tiles_per_row, int brows, int bpitch)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *As = lds;                                   // [NB * 32][kWgPitchA]
    float *Bs = lds + NB * 32 * kWgPitchA;             // [brows][bpitch]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int pl = lane & 31, h = lane >> 5;
    const int p = ks / 2;
    const int cbg = blockIdx.x % ncbg, sl = blockIdx.x / ncbg;
    const int cb = cbg * kWavesPerGroup + wave;
    const long long ntiles = (long long)B * H * tiles_per_row;
    const long long t0 = (long long)sl * tiles_per_slice;
    const long long t1 = t0 + tiles_per_slice < ntiles ? t0 + tiles_per_slice : ntiles;
    // the activation rows (i, ky) = divmod(rowid, ks) this workgroup's columns read: rowid in [r0, r0 + brows)
    const int nrow_all = L * ks;
    const int r0 = (cbg * kWavesPerGroup * 32) / ks;
    // this lane's column
    int kk = cb * 32 + pl;
    if (kk >= KK) kk = KK - 1;                          // (padded columns: any valid one, their sums are dropped)
    const int boff = (kk / ks - r0) * bpitch + kk % ks + 32 * h;
    // cooperative loads: As in float4 pieces (row, 16 per row), Bs likewise (row, bpitch / 4 per row)
    constexpr int APT = (NB * 32 * 16 + 64 * kWavesPerGroup - 1) / (64 * kWavesPerGroup);
    const int bq = bpitch / 4;
    const int nbv = brows * bq;                         // float4 pieces of Bs
    constexpr int BPT = 4;                              // prefetched pieces per thread; windows under 3 taps need more and load in place
    const bool direct = nbv > BPT * 64 * kWavesPerGroup;
    f4u ra[APT], rb[BPT];
    auto fetch = [&](long long tile) {
        const int tx = (int)(tile % tiles_per_row);
        const long long ry = tile / tiles_per_row;
        const int y = (int)(ry % H), b = (int)(ry / H);
        const int x0 = tx * 64;
        const float *gb = gpad + (size_t)b * LM * gplane + (size_t)(y + p) * WpG + x0 + p;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int idx = tid + i * 64 * kWavesPerGroup;
            const int row = idx >> 4, c4 = idx & 15;
            ra[i] = row < LM ? *reinterpret_cast<const f4u *>(gb + (size_t)row * gplane + 4 * c4) : f4u{0.f, 0.f, 0.f, 0.f};
        }
        const float *ab = apad + (size_t)b * L * plane + (size_t)y * Wp + x0;
        if (direct) return;
#pragma unroll
        for (int i = 0; i < BPT; ++i) {
            const int idx = tid + i * 64 * kWavesPerGroup;
            rb[i] = f4u{0.f, 0.f, 0.f, 0.f};
            if (idx < nbv) {
                const int row = idx / bq, c4 = idx - row * bq;
                int rid = r0 + row;
                if (rid >= nrow_all) rid = nrow_all - 1;
                const int ci = rid / ks, ky = rid - ci * ks;
                rb[i] = *reinterpret_cast<const f4u *>(ab + (size_t)ci * plane + (size_t)ky * Wp + 4 * c4);
            }
        }
    };
    auto stage = [&](long long tile) {
        if (direct) {
            const int tx = (int)(tile % tiles_per_row);
            const long long ry = tile / tiles_per_row;
            const int y = (int)(ry % H), b = (int)(ry / H);
            const float *ab = apad + (size_t)b * L * plane + (size_t)y * Wp + tx * 64;
            for (int idx = tid; idx < nbv; idx += 64 * kWavesPerGroup) {
                const int row = idx / bq, c4 = idx - row * bq;
                int rid = r0 + row;
                if (rid >= nrow_all) rid = nrow_all - 1;
                const int ci = rid / ks, ky = rid - ci * ks;
                *reinterpret_cast<f4a *>(Bs + row * bpitch + 4 * c4) =
                    *reinterpret_cast<const f4u *>(ab + (size_t)ci * plane + (size_t)ky * Wp + 4 * c4);
            }
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int idx = tid + i * 64 * kWavesPerGroup;
            const int row = idx >> 4, c4 = idx & 15;
            if (row < NB * 32) *reinterpret_cast<f4a *>(As + row * kWgPitchA + 4 * c4) = ra[i];
        }
        if (!direct) {
#pragma unroll
            for (int i = 0; i < BPT; ++i) {
                const int idx = tid + i * 64 * kWavesPerGroup;
                if (idx < nbv) {
                    const int row = idx / bq, c4 = idx - row * bq;
                    *reinterpret_cast<f4a *>(Bs + row * bpitch + 4 * c4) = rb[i];
                }
            }
        }
    };
    f16v d[NB];
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) d[m][v] = 0.f;
    if (t0 < t1) fetch(t0);
    for (long long tile = t0; tile < t1; ++tile) {
        __syncthreads();                                // the previous tile's readers are done
        stage(tile);
        __syncthreads();
        if (tile + 1 < t1) fetch(tile + 1);             // in flight while this tile is multiplied
        const float *arow = As + pl * kWgPitchA + 32 * h;
        const float *brow = Bs + boff;
#pragma unroll 2
        for (int j = 0; j < 32; j += 4) {
            float bv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = brow[j + q];
            f4a av[NB];
#pragma unroll
            for (int m = 0; m < NB; ++m) av[m] = *reinterpret_cast<const f4a *>(arow + m * 32 * kWgPitchA + j);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int m = 0; m < NB; ++m) d[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][q], bv[q], d[m], 0, 0, 0);
        }
    }
    if (cb >= ncb) return;
    // register v of lane (p, h): row 8 (v/4) + 4 h + v%4, column p
    float *out = part + ((size_t)sl * ncb + cb) * (size_t)(NB * 1024);
#pragma unroll
    for (int m = 0; m < NB; ++m)
#pragma unroll
        for (int v = 0; v < 16; ++v) out[m * 1024 + (8 * (v / 4) + 4 * h + (v % 4)) * 32 + pl] = d[m][v];
}

// d conv_w[oc][kk] = sum over the slices (in order) of part[slice][kk / 32][oc / 32][oc % 32][kk % 32]  (NB blocks of rows)
__global__ void __launch_bounds__(256)
context_wgrad_reduce_kernel(const float *__restrict__ part, float *__restrict__ d_conv_w, int LM, int NB, int KK, int ncb, int nsl)
{
    const size_t n = (size_t)LM * KK;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    const int kk = (int)(idx % KK), oc = (int)(idx / KK);
    const size_t at = ((size_t)(kk >> 5) * NB + (oc >> 5)) * 1024 + (oc & 31) * 32 + (kk & 31);
    float s = 0.f;
    for (int sl = 0; sl < nsl; ++sl) s += part[(size_t)sl * ncb * (NB * 1024) + at];
    d_conv_w[idx] = s;
}

struct ContextWorkspace {
    size_t apad, w2, koff, total;
    int Hp, Wp, KK, KKp;
    // backward only
    size_t logq, amin, cmin, gpad, wt, da, part;
    int HpG, WpG, KK2, KK2p, ncb, nsl, tiles_per_slice;
};

constexpr int kWgradWaves = 6144;     // waves of the weight-gradient launch (column blocks x pixel slices)

ContextWorkspace context_carve(int B, int L, int M, int H, int W, int ks, bool backward = false)
{
    ContextWorkspace w{};
    w.Hp = H + ks - 1;
    w.Wp = W + ks - 1;
    w.KK = L * ks * ks;
    w.KKp = (w.KK + kChunk - 1) / kChunk * kChunk;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.apad = take(((size_t)B * L * w.Hp * w.Wp + 128) * 4);
    w.w2 = take((size_t)w.KKp * 32 * M * 4);
    w.koff = take((size_t)w.KKp * 4);
    if (backward) {
        const size_t n = (size_t)B * L * H * W;
        w.logq = take(n * 4);
        w.amin = take(n);
        w.cmin = take(n * 4);
        w.HpG = H + ks - 1;
        w.WpG = (W + 63) / 64 * 64 + ks - 1;           // the last 64-pixel tile of a row ends inside the zero padding
        w.gpad = take(((size_t)B * L * M * w.HpG * w.WpG + 128) * 4);
        w.KK2 = L * M * ks * ks;
        w.KK2p = (w.KK2 + kChunk - 1) / kChunk * kChunk;
        w.wt = take((size_t)w.KK2p * 32 * 4);
        w.da = take(n * 4);
        w.ncb = w.KKp / 32;
        const long long ntiles = (long long)B * H * ((W + 63) / 64);
        long long nsl = kWgradWaves / ((w.ncb + kWavesPerGroup - 1) / kWavesPerGroup * kWavesPerGroup);
        if (nsl < 1) nsl = 1;
        if (nsl > ntiles) nsl = ntiles;
        w.tiles_per_slice = (int)((ntiles + nsl - 1) / nsl);
        w.nsl = (int)((ntiles + w.tiles_per_slice - 1) / w.tiles_per_slice);
        w.part = take((size_t)w.nsl * w.ncb * M * 1024 * 4 + (size_t)B * L * M * 4);
    }
    w.total = off;
    return w;
}

bool context_supported(int L, int M, int ks) { return L >= 1 && L <= 32 && M >= 1 && M <= 5 && ks >= 1 && ks <= 15 && (ks & 1); }

// Launch of the implicit-GEMM kernel of the forward (MODE 0: rows 32 m + l; MODE 2: packed rows).
template <int M, int MODE>
int launch(const float *planes, const float *w2, const int *koff, int KKp, int Cin, int Wp, int plane, const float *conv_b,
           const float *ga_w, const float *ga_b, const float *unary, float *smooth, float *logq, unsigned char *amin, float *cmin,
           int B, int L, int H, int W, hipStream_t st, int mixtures = M)
{
    const int tiles_per_row = (W + 31) / 32;
    const long long ntiles = (long long)B * H * tiles_per_row;
    long long groups = (ntiles + kWavesPerGroup * kTilesPerWave - 1) / (kWavesPerGroup * kTilesPerWave);
    groups = (groups + 7) / 8 * 8;                      // (the kernel deals contiguous ranges to the 8 XCDs)
    if (groups >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    // two weight stages and the tap offsets (MODE 2: or, if larger, one [32 M][32] tile per wave for the epilogue's
    // regrouping of the rows, which reuses that space)
    size_t lds = (size_t)2 * kChunk * 32 * M * 4 + (size_t)KKp * 4;
    if (MODE == 2 && lds < (size_t)kWavesPerGroup * 32 * M * 32 * 4) lds = (size_t)kWavesPerGroup * 32 * M * 32 * 4;
    if (lds > 160 * 1024) return MFL_ERR_INVALID_ARGUMENT;
    // (the attribute belongs to the current device's copy of the kernel: set before every launch)
    MFL_CHECK_HIP(hipFuncSetAttribute((const void *)label_context_kernel<M, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((label_context_kernel<M, MODE>), dim3((unsigned)groups), dim3(64 * kWavesPerGroup), lds, st, planes, w2, koff,
                       conv_b, ga_w, ga_b, unary, smooth, logq, amin, cmin, B, L, Cin, H, W, Wp, plane, KKp, tiles_per_row, mixtures);
    return (int)hipGetLastError();
}

// The activations a = la_w * t + la_b into the workspace's zero-padded planes.
int context_activations(const ContextWorkspace &ws, char *base, const float *t, const float *la_w, const float *la_b, int B, int L,
                        int H, int W, int ks, hipStream_t st)
{
    float *apad = (float *)(base + ws.apad);
    const size_t npad = (size_t)B * L * ws.Hp * ws.Wp + 128;
    MFL_CHECK_HIP(hipMemsetAsync(apad, 0, npad * 4, st));
    const size_t n = (size_t)B * L * H * W;
    const unsigned g1 = (unsigned)((n + 255) / 256 < 65535u * 8u ? (n + 255) / 256 : 65535u * 8u);
    hipLaunchKernelGGL(label_context_activation_kernel, dim3(g1), dim3(256), 0, st, t, la_w, la_b, apad, B, L, H, W, ws.Hp, ws.Wp,
                       ks / 2);
    return (int)hipGetLastError();
}

// The forward on a carved workspace: activations into padded planes, weights permuted, fused kernel.
int context_forward(const ContextWorkspace &ws, char *base, const float *t, const float *la_w, const float *la_b, const float *conv_w,
                    const float *conv_b, const float *ga_w, const float *ga_b, const float *unary, float *smooth, float *logq,
                    unsigned char *amin, float *cmin, int B, int L, int M, int H, int W, int ks, hipStream_t st)
{
    float *apad = (float *)(base + ws.apad);
    {
        const int rc = context_activations(ws, base, t, la_w, la_b, B, L, H, W, ks, st);
        if (rc) return rc;
    }
    // rows of the GEMM: 32 per mixture (the min over the mixtures is then register-wise), or, where that saves a block of
    // 32 rows, the output channels packed (21 labels x 5 mixtures: 4 blocks instead of 5)
    const int nblk = (L * M + 31) / 32;
    const bool packed = nblk < M;
    const int R = 32 * (packed ? nblk : M);
    const size_t nw = (size_t)ws.KKp * R;
    hipLaunchKernelGGL(label_context_weights_kernel, dim3((unsigned)((nw + 255) / 256)), dim3(256), 0, st, conv_w,
                       (float *)(base + ws.w2), (int *)(base + ws.koff), L, M, ks, ws.KK, ws.KKp, ws.Hp * ws.Wp, ws.Wp, R, packed ? 1 : 0);
    MFL_CHECK_HIP(hipGetLastError());
    const float *w2 = (const float *)(base + ws.w2);
    const int *koff = (const int *)(base + ws.koff);
    const int plane = ws.Hp * ws.Wp;
    if (packed) {
#define MFL_CTX_PACKED(NB) \
    case NB: return launch<NB, 2>(apad, w2, koff, ws.KKp, L, ws.Wp, plane, conv_b, ga_w, ga_b, unary, smooth, logq, amin, cmin, B, L, H, W, st, M);
        switch (nblk) {
            MFL_CTX_PACKED(1) MFL_CTX_PACKED(2) MFL_CTX_PACKED(3) MFL_CTX_PACKED(4)
        default: return MFL_ERR_INVALID_ARGUMENT;
        }
#undef MFL_CTX_PACKED
    }
#define MFL_CTX_FWD(MV) \
    case MV: return launch<MV, 0>(apad, w2, koff, ws.KKp, L, ws.Wp, plane, conv_b, ga_w, ga_b, unary, smooth, logq, amin, cmin, B, L, H, W, st);
    switch (M) {
        MFL_CTX_FWD(1) MFL_CTX_FWD(2) MFL_CTX_FWD(3) MFL_CTX_FWD(4) MFL_CTX_FWD(5)
    default: return MFL_ERR_INVALID_ARGUMENT;
    }
#undef MFL_CTX_FWD
}

template <int NB>
int launch_wgrad(const ContextWorkspace &ws, char *base, int B, int L, int LM, int H, int W, int ks, hipStream_t st)
{
    const int ncbg = (ws.ncb + kWavesPerGroup - 1) / kWavesPerGroup;
    const int brows = (kWavesPerGroup * 32) / ks + 2;
    const int bpitch = (64 + ks - 1 + 3) / 4 * 4 + 4;
    const size_t lds = ((size_t)NB * 32 * kWgPitchA + (size_t)brows * bpitch) * 4;
    MFL_CHECK_HIP(hipFuncSetAttribute((const void *)context_wgrad_kernel<NB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL(context_wgrad_kernel<NB>, dim3((unsigned)((long long)ncbg * ws.nsl)), dim3(64 * kWavesPerGroup), lds, st,
                       (const float *)(base + ws.gpad), (const float *)(base + ws.apad), (float *)(base + ws.part), B, L, LM, H, W, ks,
                       ws.Wp, ws.Hp * ws.Wp, ws.WpG, ws.HpG * ws.WpG, ws.KK, ws.ncb, ncbg, ws.nsl, ws.tiles_per_slice, (W + 63) / 64,
                       brows, bpitch);
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

size_t label_context_backward_workspace_bytes(int B, int L, int M, int H, int W, int ks)
{
    if (B <= 0 || H <= 0 || W <= 0 || !context_supported(L, M, ks)) return 0;
    if ((long long)B * L * M * (H + ks - 1) * ((W + 63) / 64 * 64 + ks - 1) >= (1LL << 31)) return 0;
    return context_carve(B, L, M, H, W, ks, true).total;
}

}  // namespace mfl

extern "C" size_t mfl_label_context_workspace_bytes(int B, int L, int M, int H, int W, int ks)
{
    return mfl::label_context_workspace_bytes(B, L, M, H, W, ks);
}

extern "C" size_t mfl_label_context_backward_workspace_bytes(int B, int L, int M, int H, int W, int ks)
{
    return mfl::label_context_backward_workspace_bytes(B, L, M, H, W, ks);
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
    return context_forward(ws, (char *)workspace, t, la_w, la_b, conv_w, conv_b, ga_w, ga_b, unary, smooth, logq, nullptr, nullptr, B, L,
                           M, H, W, ks, (hipStream_t)stream);
}

extern "C" int mfl_label_context_forward_keep(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                              const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                              float *smooth, float *logq, unsigned char *kept_amin, float *kept_cmin, int B, int L,
                                              int M, int H, int W, int ks, void *workspace, size_t workspace_bytes,
                                              mfl_stream_t stream)
{
    if (!t || !conv_w || !unary || !logq || !kept_amin || !kept_cmin) return MFL_ERR_INVALID_ARGUMENT;
    if ((la_w == nullptr) != (la_b == nullptr) || (ga_w == nullptr) != (ga_b == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || H <= 0 || W <= 0 || !context_supported(L, M, ks)) return MFL_ERR_INVALID_ARGUMENT;
    if ((long long)B * L * (H + ks - 1) * (W + ks - 1) >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const ContextWorkspace ws = context_carve(B, L, M, H, W, ks);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    return context_forward(ws, (char *)workspace, t, la_w, la_b, conv_w, conv_b, ga_w, ga_b, unary, smooth, logq, kept_amin, kept_cmin,
                           B, L, M, H, W, ks, (hipStream_t)stream);
}

namespace {

// kept_*: logq, the arg-min mixture and the convolution's value there as mfl_label_context_forward_keep wrote them, or all
// NULL: the forward is run again into the workspace.
int context_backward(const float *t, const float *la_w, const float *la_b, const float *conv_w, const float *conv_b,
                     const float *ga_w, const float *ga_b, const float *unary, const float *kept_logq,
                     const unsigned char *kept_amin, const float *kept_cmin, const float *grad_logq, const float *grad_smooth,
                     float *d_t, float *d_unary, float *d_la_w, float *d_la_b, float *d_conv_w, float *d_conv_b, float *d_ga_w,
                     float *d_ga_b, int B, int L, int M, int H, int W, int ks, void *workspace, size_t workspace_bytes,
                     mfl_stream_t stream)
{
    if (!t || !conv_w || !unary || (!grad_logq && !grad_smooth) || !d_t || !d_unary || !d_conv_w) return MFL_ERR_INVALID_ARGUMENT;
    if ((la_w == nullptr) != (la_b == nullptr) || (ga_w == nullptr) != (ga_b == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((la_w != nullptr) != (d_la_w != nullptr) || (la_w != nullptr) != (d_la_b != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((ga_w != nullptr) != (d_ga_w != nullptr) || (ga_w != nullptr) != (d_ga_b != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((conv_b != nullptr) != (d_conv_b != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || H <= 0 || W <= 0 || !context_supported(L, M, ks)) return MFL_ERR_INVALID_ARGUMENT;
    if (mfl::label_context_backward_workspace_bytes(B, L, M, H, W, ks) == 0) return MFL_ERR_TOO_LARGE;
    const ContextWorkspace ws = context_carve(B, L, M, H, W, ks, true);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    const int HW = H * W, p = ks / 2, LM = L * M;
    const size_t gplane = (size_t)ws.HpG * ws.WpG;
    float *gpad = (float *)(base + ws.gpad);
    float *da = (float *)(base + ws.da);
    const float *logq = kept_logq, *cmin = kept_cmin;
    const unsigned char *amin = kept_amin;
    int rc;
    if (kept_logq) {
        // the caller kept what the forward produced: only the padded activations (the weight gradient's operand) are rebuilt
        rc = context_activations(ws, base, t, la_w, la_b, B, L, H, W, ks, st);
    } else {
        // the forward again, keeping the arg-min mixture and the convolution's value there
        float *lq = (float *)(base + ws.logq), *cm = (float *)(base + ws.cmin);
        unsigned char *am = (unsigned char *)(base + ws.amin);
        rc = context_forward(ws, base, t, la_w, la_b, conv_w, conv_b, ga_w, ga_b, unary, nullptr, lq, am, cm, B, L, M, H, W, ks, st);
        logq = lq; cmin = cm; amin = am;
    }
    if (rc) return rc;
    MFL_CHECK_HIP(hipMemsetAsync(gpad, 0, ((size_t)B * LM * gplane + 128) * 4, st));
    if (ga_w) {
        MFL_CHECK_HIP(hipMemsetAsync(d_ga_w, 0, (size_t)LM * HW * 4, st));
        MFL_CHECK_HIP(hipMemsetAsync(d_ga_b, 0, (size_t)LM * HW * 4, st));
    }
#define MFL_CTX_PRE(LPV)                                                                                                             \
    case LPV:                                                                                                                        \
        hipLaunchKernelGGL(context_pre_kernel<LPV>, dim3((unsigned)((HW + 255) / 256)), dim3(256), 0, st, grad_logq, grad_smooth, logq, \
                           unary, amin, cmin, ga_w, gpad, d_unary, d_ga_w, d_ga_b, B, L, M, H, W, ws.HpG, ws.WpG, p);                  \
        break;
    switch ((L + 7) / 8 * 8) {
        MFL_CTX_PRE(8) MFL_CTX_PRE(16) MFL_CTX_PRE(24) MFL_CTX_PRE(32)
    }
#undef MFL_CTX_PRE
    if (d_conv_b) {
        // (the per-image sums borrow the start of the weight gradient's partial-sum buffer, which is written later)
        float *bpart = (float *)(base + ws.part);
        hipLaunchKernelGGL(context_bias_grad_kernel, dim3((unsigned)LM, (unsigned)B), dim3(256), 0, st, (const float *)gpad, bpart, LM, gplane);
        hipLaunchKernelGGL(context_bias_grad_sum_kernel, dim3((unsigned)((LM + 255) / 256)), dim3(256), 0, st, (const float *)bpart, d_conv_b, B, LM);
    }
    // data gradient: one block of 32 rows over the L*M gradient planes
    {
        const size_t nw = (size_t)ws.KK2p * 32;
        hipLaunchKernelGGL(context_dgrad_weights_kernel, dim3((unsigned)((nw + 255) / 256 < 65535u ? (nw + 255) / 256 : 65535u)), dim3(256), 0, st,
                           conv_w, (float *)(base + ws.wt), L, LM, ks, ws.KK2, ws.KK2p);
        MFL_CHECK_HIP(hipGetLastError());
        const float *wtp = (const float *)(base + ws.wt);
        float *dout = la_w ? da : d_t;
#define MFL_CTX_DGRAD(KSV) case KSV: rc = launch_dgrad<KSV>(gpad, wtp, dout, B, L, LM, H, W, ws.WpG, gplane, st); break;
        switch (ks) {
            MFL_CTX_DGRAD(1) MFL_CTX_DGRAD(3) MFL_CTX_DGRAD(5) MFL_CTX_DGRAD(7) MFL_CTX_DGRAD(9) MFL_CTX_DGRAD(11) MFL_CTX_DGRAD(13)
            MFL_CTX_DGRAD(15)
        default: rc = MFL_ERR_INVALID_ARGUMENT;
        }
#undef MFL_CTX_DGRAD
        if (rc) return rc;
        if (la_w) {
            const size_t LHW = (size_t)L * HW;
            hipLaunchKernelGGL(context_affine_grad_kernel, dim3((unsigned)((LHW + 255) / 256)), dim3(256), 0, st, (const float *)da, t, la_w,
                               d_t, d_la_w, d_la_b, B, LHW);
        }
    }
    // weight gradient
    const int nb = (LM + 31) / 32;               // blocks of 32 rows (output channels packed)
    switch (nb) {
    case 1: rc = launch_wgrad<1>(ws, base, B, L, LM, H, W, ks, st); break;
    case 2: rc = launch_wgrad<2>(ws, base, B, L, LM, H, W, ks, st); break;
    case 3: rc = launch_wgrad<3>(ws, base, B, L, LM, H, W, ks, st); break;
    case 4: rc = launch_wgrad<4>(ws, base, B, L, LM, H, W, ks, st); break;
    default: rc = launch_wgrad<5>(ws, base, B, L, LM, H, W, ks, st); break;
    }
    if (rc) return rc;
    {
        const size_t n = (size_t)LM * ws.KK;
        hipLaunchKernelGGL(context_wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float *)(base + ws.part),
                           d_conv_w, LM, nb, ws.KK, ws.ncb, ws.nsl);
    }
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mfl_label_context_backward(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                          const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                          const float *grad_logq, const float *grad_smooth, float *d_t, float *d_unary,
                                          float *d_la_w, float *d_la_b, float *d_conv_w, float *d_conv_b, float *d_ga_w,
                                          float *d_ga_b, int B, int L, int M, int H, int W, int ks, void *workspace,
                                          size_t workspace_bytes, mfl_stream_t stream)
{
    return context_backward(t, la_w, la_b, conv_w, conv_b, ga_w, ga_b, unary, nullptr, nullptr, nullptr, grad_logq, grad_smooth, d_t,
                            d_unary, d_la_w, d_la_b, d_conv_w, d_conv_b, d_ga_w, d_ga_b, B, L, M, H, W, ks, workspace, workspace_bytes,
                            stream);
}

extern "C" int mfl_label_context_backward_kept(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                               const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                               const float *kept_logq, const unsigned char *kept_amin, const float *kept_cmin,
                                               const float *grad_logq, const float *grad_smooth, float *d_t, float *d_unary,
                                               float *d_la_w, float *d_la_b, float *d_conv_w, float *d_conv_b, float *d_ga_w,
                                               float *d_ga_b, int B, int L, int M, int H, int W, int ks, void *workspace,
                                               size_t workspace_bytes, mfl_stream_t stream)
{
    if (!kept_logq || !kept_amin || !kept_cmin) return MFL_ERR_INVALID_ARGUMENT;
    return context_backward(t, la_w, la_b, conv_w, conv_b, ga_w, ga_b, unary, kept_logq, kept_amin, kept_cmin, grad_logq, grad_smooth,
                            d_t, d_unary, d_la_w, d_la_b, d_conv_w, d_conv_b, d_ga_w, d_ga_b, B, L, M, H, W, ks, workspace,
                            workspace_bytes, stream);
}
