// Backward of the K-iteration mean-field loop (C ABI entry point mfl_meanfield_backward): what autograd
// derives when loss.backward() (AdvancedModel.py:299) runs through the composition
//   M_t = Conv2dLocal(Q_t)                LocalConvolution.py:59-90
//   A_t = aw * M_t + ab                    AdvancedModel.py:52-56, 223
//   P_t = compat . A_t                     AdvancedModel.py:143-144 with a 1x1 kernel
//   logQ_{t+1} = LogSoftmax(log_softmax(U) - P_t),  Q_{t+1} = exp(logQ_{t+1}),  Q_0 = U    AdvancedModel.py:222, 242-244
// iterated K times.
//
// Scheme: recompute with checkpoints.  The forward loop runs once more with logQ_{t+1} of every iteration
// kept; then, for t = K-1 .. 0:
//   X_t  = Q_t (t = 0: U; else exp of the kept logQ_t)                         exp kernel
//   M_t  = local forward of X_t                                                mfl_local_forward
//   pre  : g = incoming gradient of logQ_{t+1} (t = K-1: the caller's; else dQ_{t+1} * Q_{t+1})
//          dz = g - Q_{t+1} * sum_c g;  dls += dz;  dP = -dz;  dA = compat^T . dP;  dM = aw * dA
//          daw += sum_b dA * M_t;  dab += sum_b dA;  dcompat += sum_{b,l} dP (x) A_t    loop_pre_kernel + reduce
//   (dQ_t, d pos_w, d col_w) = local backward of (X_t, dM)                     mfl_local_backward
// and finally dU = dQ_0 + dls - softmax(U) * sum_c dls.
// Every sum over pixels or images has a fixed order (no atomics here): apart from the scatter inside
// mfl_local_backward the result is reproducible bit for bit.
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

constexpr int kPix = 64;          // pixels per workgroup of the pre kernel (one wave)
constexpr int kPitch = kPix + 1;  // LDS pitch of a channel row: rows read across channels stay conflict-free
constexpr int kMaxC = 32;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ void loop_exp_kernel(const float *__restrict__ logq, float *__restrict__ q, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        q[i] = __expf(logq[i]);
}

typedef const float __attribute__((address_space(4))) *ConstFloats;   // scalar loads: never written while a kernel reads it
typedef float f16v __attribute__((ext_vector_type(16)));

// ct32[j][i] = compat[i][j], zero-padded to 32 x 32: row j holds what dA_j = sum_i compat[i][j] * dP_i needs,
// 128 contiguous bytes that the pre kernel reads with scalar loads.
__global__ void loop_pad_compat_kernel(const float *__restrict__ compat, float *__restrict__ ct32, int C)
{
    const int j = threadIdx.x >> 5, i = threadIdx.x & 31;
    ct32[threadIdx.x] = (i < C && j < C) ? compat[i * C + j] : 0.f;
}

// One workgroup (one wave) per 64 consecutive pixels of the image plane, looping over the batch so that
// the sums over images (daw, dab) and over the pixels of the tile (dcompat) have a fixed order.  A lane owns one
// pixel and keeps its channel vectors in registers (CE = label count rounded up to 8; channels past C hold zeros):
// the label product dA = compat^T . dP takes the matrix rows from scalar registers, and the tile's share of
// dcompat = sum_px dP (x) A runs on the matrix cores: 32 steps of v_mfma_f32_32x32x2_f32 (fp32 in and out) over the
// tile's 64 pixels, rows = dP channels, columns = A channels, accumulated over the images in the same registers.
//   g_in:    t = K-1: dL/dlogQ_K of the caller;  else dQ_{t+1} of the last local backward (from_dq)
//   logq1:   logQ_{t+1}
//   m:       M_t (local forward of Q_t), or NULL: then M_t = Q_t * (pos_w * P + col_w * Q) from the plain window
//            sums P, Q of the local backward's first phase (xq = Q_t)
//   ct32:    the padded transposed label matrix (MAT)
//   dm:      out, gradient of M_t
//   dls:     running sum of dz (first: overwritten)
//   daw/dab: running sums (first: overwritten), [C][L] (AFF)
//   part:    out, this workgroup's C x C partial of dcompat (MAT)
template <int CE, bool AFF, bool MAT>
__global__ void __launch_bounds__(kPix)
loop_pre_kernel(const float *__restrict__ g_in, int from_dq, const float *__restrict__ logq1, const float *__restrict__ m,
                const float *__restrict__ xq, const float *__restrict__ psum, const float *__restrict__ qsum,
                const float *__restrict__ pos_w, const float *__restrict__ col_w, const float *__restrict__ aw, const float *__restrict__ ab,
                const float *__restrict__ ct32, float *__restrict__ dm, float *__restrict__ dls, float *__restrict__ daw,
                float *__restrict__ dab, float *__restrict__ part, int first, int B, int C, int L)
{
    extern __shared__ float lds[];
    float *dp = lds;                       // [32][kPitch]  dP of the current image (MAT)
    float *at = dp + 32 * kPitch;          // [32][kPitch]  A_t of the current image (MAT)
    const int lane = threadIdx.x;
    const int l0 = blockIdx.x * kPix + lane;
    const bool ok = l0 < L;
    const int l = ok ? l0 : L - 1;         // (lanes past the plane load the last pixel and keep zeros)
    const float pw = m ? 0.f : *pos_w, cw = m ? 0.f : *col_w;
    if constexpr (MAT) {
        for (int i = lane; i < 2 * 32 * kPitch; i += kPix) lds[i] = 0.f;     // rows of the channels past CE stay zero
        __syncthreads();
    }
    float sw[CE], sb[CE];
#pragma unroll
    for (int c = 0; c < CE; ++c) { sw[c] = 0.f; sb[c] = 0.f; }
    f16v dacc;
#pragma unroll
    for (int v = 0; v < 16; ++v) dacc[v] = 0.f;
    const int p = lane & 31, h = lane >> 5;
    for (int b = 0; b < B; ++b) {
        const size_t base = (size_t)b * C * L + l;
        // (everything of a pixel stays in registers and all loads of an image are requested at once: measured faster
        // than parking the affine values in LDS or bounding the loads in flight, which allow more waves per SIMD)
        float g[CE], q[CE], a[CE], mv[CE];
        float sg = 0.f;
#pragma unroll
        for (int c = 0; c < CE; ++c) {
            const size_t o = base + (size_t)(c < C ? c : C - 1) * L;
            const bool live = ok && c < C;
            q[c] = __expf(logq1[o]);
            float gv = g_in[o];
            if (from_dq) gv *= q[c];
            g[c] = live ? gv : 0.f;
            mv[c] = m ? m[o] : xq[o] * fmaf(pw, psum[o], cw * qsum[o]);
            float av = mv[c];
            if constexpr (AFF) {
                const size_t oc = (size_t)(c < C ? c : C - 1) * L + l;
                av = fmaf(aw[oc], av, ab[oc]);
            }
            a[c] = live ? av : 0.f;
            sg += g[c];
        }
        // dz = g - Q * sum g;  dP = -dz (kept in g)
#pragma unroll
        for (int c = 0; c < CE; ++c) {
            const float dz = g[c] - q[c] * sg;
            if (ok && c < C) {
                const size_t o = base + (size_t)c * L;
                dls[o] = first ? dz : dls[o] + dz;
            }
            g[c] = (ok && c < C) ? -dz : 0.f;
        }
        // dA_j = sum_i compat[i][j] * dP_i;  dM_j = aw_j * dA_j
#pragma unroll
        for (int j = 0; j < CE; ++j) {
            float da;
            if constexpr (MAT) {
                ConstFloats row = (ConstFloats)(uintptr_t)(ct32 + j * 32);
                da = 0.f;
#pragma unroll
                for (int i = 0; i < CE; ++i) da = fmaf(row[i], g[i], da);
            } else {
                da = g[j];
            }
            float dmv = da;
            if constexpr (AFF) {
                sw[j] = fmaf(da, mv[j], sw[j]);
                sb[j] += da;
                dmv = aw[(size_t)(j < C ? j : C - 1) * L + l] * da;
            }
            if (ok && j < C) dm[base + (size_t)j * L] = dmv;
        }
        if constexpr (MAT) {
            // the tile's dP and A through LDS, then D[i][j] += sum_px dP_i(px) * A_j(px): lane (p, h) feeds row / column p
            // of pixel 2 s + h at step s and holds D[8 (v / 4) + 4 h + v % 4][p] in register v
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                dp[c * kPitch + lane] = g[c];
                at[c * kPitch + lane] = a[c];
            }
            __syncthreads();
#pragma unroll 8
            for (int s = 0; s < kPix / 2; ++s)
                dacc = __builtin_amdgcn_mfma_f32_32x32x2f32(dp[p * kPitch + 2 * s + h], at[p * kPitch + 2 * s + h], dacc, 0, 0, 0);
            __syncthreads();
        }
    }
    if constexpr (AFF) {
        if (ok) {
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                if (c < C) {
                    const size_t o = (size_t)c * L + l;
                    daw[o] = first ? sw[c] : daw[o] + sw[c];
                    dab[o] = first ? sb[c] : dab[o] + sb[c];
                }
            }
        }
    }
    if constexpr (MAT) {
#pragma unroll
        for (int v = 0; v < 16; ++v) {
            const int i = 8 * (v / 4) + 4 * h + (v % 4);
            if (i < C && p < C) part[(size_t)blockIdx.x * C * C + i * C + p] = dacc[v];
        }
    }
}

// dcompat[i][j] (+)= sum over the workgroups' partials: one workgroup per matrix element, every thread sums a
// fixed strided subset of the partials, then a fixed tree in LDS (the same order in every run).
__global__ void __launch_bounds__(256)
loop_compat_reduce_kernel(const float *__restrict__ part, int nblk, int CC, float *__restrict__ dcompat, int first)
{
    __shared__ float sh[256];
    const int idx = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(size_t)b * CC + idx];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) dcompat[idx] = first ? sh[0] : dcompat[idx] + sh[0];
}

// dU = dQ_0 + dls - softmax(U) * sum_c dls; one thread per pixel.  dparams = sum of the per-iteration pairs.
__global__ void loop_final_kernel(const float *__restrict__ unary, const float *__restrict__ dq0, const float *__restrict__ dls,
                                  float *__restrict__ du, const float *__restrict__ dpar_slots, float *__restrict__ dparams,
                                  int iters, int B, int C, int L)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        float a = 0.f, b = 0.f;
        for (int t = iters - 1; t >= 0; --t) { a += dpar_slots[2 * t]; b += dpar_slots[2 * t + 1]; }
        dparams[0] = a;
        dparams[1] = b;
    }
    if (i >= (size_t)B * L) return;
    const size_t b = i / L, l = i - b * L;
    const size_t base = b * C * L + l;
    float mx = -INFINITY, sd = 0.f;
    for (int c = 0; c < C; ++c) {
        mx = fmaxf(mx, unary[base + (size_t)c * L]);
        sd += dls[base + (size_t)c * L];
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(unary[base + (size_t)c * L] - mx);
    const float rs = 1.f / se;
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * L;
        du[o] = dq0[o] + dls[o] - __expf(unary[o] - mx) * rs * sd;
    }
}


// ------------------------------------------------------------------ more than 32 labels
// A pixel's channel vectors no longer fit one lane's registers, so the step is split: an elementwise kernel
// (softmax backward, M_t, A_t), the label product dA = compat^T . dP and the matrix gradient dcompat += dP (x) A
// as GEMMs on v_mfma_f32_32x32x2_f32 (fp32 in and out), and an elementwise kernel for the affine.  All sums over
// images and pixels have a fixed order.

// One thread per (image, pixel):  g = incoming gradient of logQ_{t+1} (from_dq: dQ_{t+1} * Q_{t+1});
//   dz = g - Q * sum_c g;  dls (+)= dz;  dp = -dz;  M_t = m_in, or xq * (pos_w * P + col_w * Q) from the plain window
//   sums;  a_out = aw * M + ab (or M).  m_out / a_out may be NULL.
__global__ void __launch_bounds__(256)
loop_softmax_grad_kernel(const float *__restrict__ g_in, int from_dq, const float *__restrict__ logq1, const float *__restrict__ m_in,
                         const float *__restrict__ xq, const float *__restrict__ psum, const float *__restrict__ qsum,
                         const float *__restrict__ pos_w, const float *__restrict__ col_w, const float *__restrict__ aw,
                         const float *__restrict__ ab, float *__restrict__ dp, float *__restrict__ a_out, float *__restrict__ m_out,
                         float *__restrict__ dls, int first, int B, int C, int L)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)B * L) return;
    const size_t b = i / L, l = i - b * L;
    const size_t base = b * C * L + l;
    const float pw = m_in ? 0.f : *pos_w, cw = m_in ? 0.f : *col_w;
    float sg = 0.f;
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * L;
        sg += from_dq ? g_in[o] * __expf(logq1[o]) : g_in[o];
    }
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * L;
        const float q = __expf(logq1[o]);
        const float gv = from_dq ? g_in[o] * q : g_in[o];
        const float dz = gv - q * sg;
        dls[o] = first ? dz : dls[o] + dz;
        dp[o] = -dz;
        if (m_out || a_out) {
            const float mv = m_in ? m_in[o] : xq[o] * fmaf(pw, psum[o], cw * qsum[o]);
            if (m_out) m_out[o] = mv;
            if (a_out) a_out[o] = aw ? fmaf(aw[(size_t)c * L + l], mv, ab[(size_t)c * L + l]) : mv;
        }
    }
}

constexpr int kMatWaves = 8;

// out[b][j][l] = sum_i mat[i][j] * in[b][i][l]  (dA = compat^T . dP): the label step's GEMM (mfl_label_step.hip) without its
// epilogue.  A wave owns tiles of 32 consecutive pixels; lane (p, h) supplies in[pixel p][channel 2s + h] per k-step
// and the matrix rows come from a zero-padded copy in LDS; register v of block mb = out channel 32 mb + 8 (v/4) + 4 h + v%4.
template <int MB>
__global__ void __launch_bounds__(64 * kMatWaves)
loop_matmul_kernel(const float *__restrict__ in, const float *__restrict__ mat, float *__restrict__ out, int B, int C, int L)
{
    extern __shared__ __attribute__((aligned(16))) float ct[];      // [2 * KJ][CP]: ct[i][j] = mat[i][j]
    constexpr int CP = 32 * MB;
    const int KJ = (C + 1) / 2;
    for (int idx = threadIdx.x; idx < 2 * KJ * CP; idx += blockDim.x) {
        const int i = idx / CP, j = idx - i * CP;
        ct[idx] = (i < C && j < C) ? mat[(size_t)i * C + j] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const long long npix = (long long)B * L;
    const long long ntiles = (npix + 31) / 32;
    for (long long tile = (long long)blockIdx.x * kMatWaves + wave; tile < ntiles; tile += (long long)gridDim.x * kMatWaves) {
        const long long pix = tile * 32 + p;
        const bool ok = pix < npix;
        const long long pc = ok ? pix : npix - 1;
        const int b = (int)(pc / L), l = (int)(pc - (long long)b * L);
        const float *src = in + ((size_t)b * C) * L + l;
        f16v d[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) d[mb][v] = 0.f;
        constexpr int kAhead = 8;
        for (int j0 = 0; j0 < KJ; j0 += kAhead) {
            float av[kAhead];
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const int c2 = 2 * (j0 + jj) + h;
                av[jj] = c2 < C ? src[(size_t)c2 * L] : 0.f;
            }
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                if (j0 + jj < KJ) {
                    const float *row = ct + (2 * (j0 + jj) + h) * CP + p;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) d[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32 * mb], av[jj], d[mb], 0, 0, 0);
                }
            }
        }
        if (ok) {
            float *dst = out + ((size_t)b * C) * L + l;
#pragma unroll
            for (int mb = 0; mb < MB; ++mb)
#pragma unroll
                for (int v = 0; v < 16; ++v) {
                    const int c = 32 * mb + 8 * (v / 4) + 4 * h + (v % 4);
                    if (c < C) dst[(size_t)c * L] = d[mb][v];
                }
        }
    }
}

// dm = aw * dA;  daw (+)= sum_b dA * M;  dab (+)= sum_b dA.  One thread per (channel, pixel), looping over the batch.
__global__ void __launch_bounds__(256)
loop_affine_grad_kernel(const float *__restrict__ da, const float *__restrict__ m, const float *__restrict__ aw, float *__restrict__ dm,
                        float *__restrict__ daw, float *__restrict__ dab, int first, int B, size_t CL)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= CL) return;
    const float w = aw[i];
    float sw = 0.f, sb = 0.f;
    for (int b = 0; b < B; ++b) {
        const float g = da[(size_t)b * CL + i];
        dm[(size_t)b * CL + i] = w * g;
        sw = fmaf(g, m[(size_t)b * CL + i], sw);
        sb += g;
    }
    daw[i] = first ? sw : daw[i] + sw;
    dab[i] = first ? sb : dab[i] + sb;
}

typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// Four consecutive floats of a plane row starting at pixel l; zeros past the plane's end (and for rows that do not exist).
__device__ __forceinline__ f4u load4_guarded(const float *row, int l, int L, bool row_ok)
{
    f4u v = {0.f, 0.f, 0.f, 0.f};
    if (!row_ok) return v;
    if (l + 3 < L) return *reinterpret_cast<const f4u *>(row + l);
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (l + q < L) v[q] = row[l + q];
    return v;
}

// part[slice][cb][rb][i][j] = sum over the slice's pixels of dP[32 rb + i][pixel] * A[32 cb + j][pixel]: a GEMM with the
// pixels as the contraction.  One wave per (column block cb, slice); a slice is a run of tiles of 64 consecutive pixels of
// one image; per matrix instruction lane (p, h) supplies row / column p at pixel 32 h + s of the tile, so every lane walks
// 32 consecutive floats of its own channel plane, four at a time.
template <int MB>
__global__ void __launch_bounds__(256)
loop_dcompat_kernel(const float *__restrict__ dp, const float *__restrict__ a, float *__restrict__ part, int B, int C, int L, int nsl,
                    int tiles_per_slice, int tiles_per_image)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const long long unit = (long long)blockIdx.x * 4 + wave;
    if (unit >= (long long)MB * nsl) return;
    const int cb = (int)(unit % MB), sl = (int)(unit / MB);
    const long long ntiles = (long long)B * tiles_per_image;
    const long long t0 = (long long)sl * tiles_per_slice;
    const long long t1 = t0 + tiles_per_slice < ntiles ? t0 + tiles_per_slice : ntiles;
    const bool col_ok = 32 * cb + p < C;
    f16v d[MB];
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) d[mb][v] = 0.f;
    for (long long tile = t0; tile < t1; ++tile) {
        const int b = (int)(tile / tiles_per_image);
        const int l0 = (int)(tile - (long long)b * tiles_per_image) * 64 + 32 * h;
        const float *acol = a + ((size_t)b * C + (col_ok ? 32 * cb + p : 0)) * L;
#pragma unroll 2
        for (int s = 0; s < 32; s += 4) {
            const f4u bv = load4_guarded(acol, l0 + s, L, col_ok);
            f4u av[MB];
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const bool row_ok = 32 * mb + p < C;
                av[mb] = load4_guarded(dp + ((size_t)b * C + (row_ok ? 32 * mb + p : 0)) * L, l0 + s, L, row_ok);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) d[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[mb][q], bv[q], d[mb], 0, 0, 0);
        }
    }
    float *out = part + ((size_t)sl * MB + cb) * (size_t)(MB * 1024);
#pragma unroll
    for (int mb = 0; mb < MB; ++mb)
#pragma unroll
        for (int v = 0; v < 16; ++v) out[mb * 1024 + (8 * (v / 4) + 4 * h + (v % 4)) * 32 + p] = d[mb][v];
}

// dcompat[i][j] (+)= sum over the slices, in order, of part[slice][j / 32][i / 32][i % 32][j % 32]
__global__ void __launch_bounds__(256)
loop_dcompat_reduce_kernel(const float *__restrict__ part, float *__restrict__ dcompat, int C, int MB, int nsl, int first)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= C * C) return;
    const int i = idx / C, j = idx - i * C;
    const size_t at = ((size_t)(j >> 5) * MB + (i >> 5)) * 1024 + (size_t)(i & 31) * 32 + (j & 31);
    float sum = 0.f;
    for (int sl = 0; sl < nsl; ++sl) sum += part[(size_t)sl * MB * MB * 1024 + at];
    dcompat[idx] = first ? sum : dcompat[idx] + sum;
}

constexpr int kMaxLargeC = 160;
constexpr int kDcompatWaves = 4096;

struct LoopWorkspace {
    size_t logq_all, x, m, dm, dx, dls, dpar, part, ct32, sub, sub_bytes, total;
    size_t dp, abuf, da;             // more than 32 labels only
    int nblk;
    int mb, nsl, tiles_per_slice, tiles_per_image;
};

LoopWorkspace loop_carve(int B, int C, int H, int W, int k, int iters)
{
    LoopWorkspace w{};
    const size_t n = (size_t)B * C * H * W * 4;
    const int L = H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.logq_all = take(n * (size_t)iters);
    w.x = take(n);
    w.m = take(n);
    w.dm = take(n);
    w.dx = take(n);
    w.dls = take(n);
    w.dpar = take((size_t)iters * 2 * 4);
    w.nblk = (L + kPix - 1) / kPix;
    if (C <= kMaxC) {
        w.part = take((size_t)w.nblk * C * C * 4);
        w.ct32 = take(32 * 32 * 4);
    } else {
        w.dp = take(n);
        w.abuf = take(n);
        w.da = take(n);
        w.mb = (C + 31) / 32;
        w.tiles_per_image = (L + 63) / 64;
        const long long ntiles = (long long)B * w.tiles_per_image;
        long long nsl = kDcompatWaves / w.mb;
        if (nsl < 1) nsl = 1;
        if (nsl > ntiles) nsl = ntiles;
        w.tiles_per_slice = (int)((ntiles + nsl - 1) / nsl);
        w.nsl = (int)((ntiles + w.tiles_per_slice - 1) / w.tiles_per_slice);
        w.part = take((size_t)w.nsl * w.mb * w.mb * 1024 * 4);
    }
    size_t sub = mfl_workspace_bytes(MFL_OP_MEANFIELD_FORWARD, B, C, H, W, k, iters);
    const size_t f = mfl_workspace_bytes(MFL_OP_LOCAL_FORWARD, B, C, H, W, k, 1);
    const size_t bw = mfl_workspace_bytes(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, 1);
    if (f > sub) sub = f;
    if (bw > sub) sub = bw;
    w.sub_bytes = sub;
    w.sub = take(sub);
    w.total = off;
    return w;
}

}  // namespace

namespace mfl {

// (mfl_api.hip: the diagonal local backward in two phases)
bool local_backward_split_supported(int B, int C, int H, int W, int k);
int local_backward_sums(const float *x, const float *img, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                        mfl_stream_t stream, const float **P, const float **Q);
int local_backward_apply(const float *x, const float *img, const float *pos_w, const float *col_w, const float *dy, float *dx,
                         float *dparams, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                         mfl_stream_t stream);

// (declared in mfl_api.hip: the forward loop that keeps every iterate)
int meanfield_forward_keep(const float *unary, const float *img, const float *pos_w, const float *col_w, const float *affine_w,
                           const float *affine_b, const float *compat, float *logq_all, int B, int C, int H, int W, int k,
                           int iters, void *workspace, size_t workspace_bytes, mfl_stream_t stream);

size_t loop_backward_workspace_bytes(int B, int C, int H, int W, int k, int iters)
{
    if (B <= 0 || C <= 0 || C > kMaxLargeC || H <= 0 || W <= 0 || k <= 0 || iters <= 0) return 0;
    if (mfl_workspace_bytes(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, 1) == 0) return 0;
    return loop_carve(B, C, H, W, k, iters).total;
}

}  // namespace mfl

extern "C" int mfl_meanfield_backward(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                      const float *affine_w, const float *affine_b, const float *compat,
                                      const float *logq_iterates, const float *grad_logq, float *d_unary, float *dparams,
                                      float *d_affine_w,
                                      float *d_affine_b, float *d_compat, int B, int C, int H, int W, int k, int iters,
                                      void *workspace, size_t workspace_bytes, mfl_stream_t stream)
{
    if (!unary || !img || !pos_w || !col_w || !grad_logq || !d_unary || !dparams) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || iters <= 0) return MFL_ERR_INVALID_ARGUMENT;
    if ((affine_w == nullptr) != (affine_b == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((affine_w != nullptr) != (d_affine_w != nullptr) || (affine_w != nullptr) != (d_affine_b != nullptr))
        return MFL_ERR_INVALID_ARGUMENT;
    if ((compat != nullptr) != (d_compat != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (C > kMaxLargeC) return MFL_ERR_INVALID_ARGUMENT;      // the label products hold the matrix in LDS (160 x 160 floats)
    if ((long long)B * C * H * W >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const size_t need = mfl::loop_backward_workspace_bytes(B, C, H, W, k, iters);
    if (need == 0) return MFL_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < need) return MFL_ERR_WORKSPACE;
    const LoopWorkspace ws = loop_carve(B, C, H, W, k, iters);
    char *base = (char *)workspace;
    const float *logq_all = logq_iterates ? logq_iterates : (const float *)(base + ws.logq_all);
    float *x = (float *)(base + ws.x), *m = (float *)(base + ws.m);
    float *dm = (float *)(base + ws.dm), *dx = (float *)(base + ws.dx), *dls = (float *)(base + ws.dls);
    float *dpar = (float *)(base + ws.dpar), *part = (float *)(base + ws.part), *ct32 = (float *)(base + ws.ct32);
    void *sub = base + ws.sub;
    hipStream_t st = (hipStream_t)stream;
    const int L = H * W;
    const size_t n = (size_t)B * C * L;

    // The nested operator calls keep their address tables in `sub` and skip rebuilding them when they find their own
    // plan tag there.  That tag sits in the middle of this call's workspace, where no caller-side rule protects it
    // (a block that went back to an allocator in between may come back with the tag intact and the tables
    // overwritten), so it is not trusted across calls: cleared here, the tables are built once per call and reused
    // by its K iterations.
    MFL_CHECK_HIP(hipMemsetAsync(sub, 0, 256, st));
    int rc = 0;
    if (!logq_iterates) {
        // no iterates from the caller (mfl_meanfield_forward_iterates): the forward loop runs again and keeps them
        rc = mfl::meanfield_forward_keep(unary, img, pos_w, col_w, affine_w, affine_b, compat, (float *)(base + ws.logq_all), B, C, H,
                                         W, k, iters, sub, ws.sub_bytes, stream);
        if (rc) return rc;
    }
    const size_t lds = compat ? (size_t)2 * 32 * kPitch * 4 : 0;
    if (compat && C <= kMaxC) hipLaunchKernelGGL(loop_pad_compat_kernel, dim3(1), dim3(1024), 0, st, compat, ct32, C);
    // the pre kernel of this call: channel capacity (C rounded up to 8) x affine x label matrix
    using PreKernel = void (*)(const float *, int, const float *, const float *, const float *, const float *, const float *,
                               const float *, const float *, const float *, const float *, const float *, float *, float *, float *,
                               float *, float *, int, int, int, int);
    PreKernel pre = nullptr;
    {
        const bool aff = affine_w != nullptr, mat = compat != nullptr;
#define MFL_PRE_PICK(CEV)                                                                     \
        pre = aff ? (mat ? loop_pre_kernel<CEV, true, true> : loop_pre_kernel<CEV, true, false>) \
                  : (mat ? loop_pre_kernel<CEV, false, true> : loop_pre_kernel<CEV, false, false>)
        if (C <= 8) MFL_PRE_PICK(8);
        else if (C <= 16) MFL_PRE_PICK(16);
        else if (C <= 24) MFL_PRE_PICK(24);
        else MFL_PRE_PICK(32);      // (more than 32 labels take the split kernels below)
#undef MFL_PRE_PICK
    }
    const bool split = mfl::local_backward_split_supported(B, C, H, W, k);
    const unsigned egrid = (unsigned)((n + 255) / 256 < 65535 * 8 ? (n + 255) / 256 : 65535 * 8);
    for (int t = iters - 1; t >= 0; --t) {
        const float *xt = unary;
        if (t > 0) {
            hipLaunchKernelGGL(loop_exp_kernel, dim3(egrid), dim3(256), 0, st, logq_all + (size_t)(t - 1) * n, x, n);
            xt = x;
        }
        // Where the diagonal kernels serve the local backward, its first phase (the plain window sums P, Q, which do
        // not depend on the incoming gradient) also yields M_t = X_t * (pos_w * P + col_w * Q): no separate local forward.
        const float *psum = nullptr, *qsum = nullptr;
        if (split) {
            rc = mfl::local_backward_sums(xt, img, B, C, H, W, k, sub, ws.sub_bytes, stream, &psum, &qsum);
        } else {
            rc = mfl_local_forward(xt, img, nullptr, pos_w, col_w, m, B, C, H, W, k, sub, ws.sub_bytes, stream);
        }
        if (rc) return rc;
        const int first = t == iters - 1;
        if (C <= kMaxC) {
            hipLaunchKernelGGL(pre, dim3(ws.nblk), dim3(kPix), lds, st, first ? grad_logq : dx, first ? 0 : 1,
                               logq_all + (size_t)t * n, split ? nullptr : m, xt, psum, qsum, pos_w, col_w, affine_w, affine_b,
                               (const float *)ct32, dm, dls, d_affine_w, d_affine_b, compat ? part : nullptr, first, B, C, L);
            if (compat)
                hipLaunchKernelGGL(loop_compat_reduce_kernel, dim3(C * C), dim3(256), 0, st, part, ws.nblk, C * C, d_compat, first);
        } else {
            // more than 32 labels: elementwise step, then the two label products as GEMMs, then the affine
            float *dp = (float *)(base + ws.dp), *abuf = (float *)(base + ws.abuf), *da = (float *)(base + ws.da);
            const bool aff = affine_w != nullptr, mat = compat != nullptr;
            // where the gradient of A_t lands: dm directly when nothing follows it
            float *dp_dst = (!mat && !aff) ? dm : (mat ? dp : da);
            float *m_dst = split && aff ? m : nullptr;                     // M_t is needed again for the affine's gradient
            hipLaunchKernelGGL(loop_softmax_grad_kernel, dim3((unsigned)(((size_t)B * L + 255) / 256)), dim3(256), 0, st,
                               first ? grad_logq : dx, first ? 0 : 1, logq_all + (size_t)t * n, split ? nullptr : m, xt, psum, qsum, pos_w,
                               col_w, affine_w, affine_b, dp_dst, mat ? abuf : nullptr, m_dst, dls, first, B, C, L);
            if (mat) {
                float *da_dst = aff ? da : dm;
                const long long ntiles = ((long long)B * L + 31) / 32;
                long long blocks = (ntiles + kMatWaves - 1) / kMatWaves;
                if (blocks > 256) blocks = 256;
                const size_t mlds = (size_t)2 * ((C + 1) / 2) * 32 * ws.mb * 4;
                const long long units = (long long)ws.mb * ws.nsl;
#define MFL_LARGE_C(MBV)                                                                                                              \
    case MBV:                                                                                                                         \
        MFL_CHECK_HIP(hipFuncSetAttribute((const void *)loop_matmul_kernel<MBV>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
        hipLaunchKernelGGL(loop_matmul_kernel<MBV>, dim3((unsigned)blocks), dim3(64 * kMatWaves), mlds, st, (const float *)dp, compat,   \
                           da_dst, B, C, L);                                                                                          \
        hipLaunchKernelGGL(loop_dcompat_kernel<MBV>, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, st, (const float *)dp,             \
                           (const float *)abuf, part, B, C, L, ws.nsl, ws.tiles_per_slice, ws.tiles_per_image);                         \
        break;
                switch (ws.mb) {
                    MFL_LARGE_C(2) MFL_LARGE_C(3) MFL_LARGE_C(4) MFL_LARGE_C(5)
                default: return MFL_ERR_INVALID_ARGUMENT;
                }
#undef MFL_LARGE_C
                hipLaunchKernelGGL(loop_dcompat_reduce_kernel, dim3((unsigned)((C * C + 255) / 256)), dim3(256), 0, st, (const float *)part,
                                   d_compat, C, ws.mb, ws.nsl, first);
            }
            if (aff) {
                const size_t CL = (size_t)C * L;
                hipLaunchKernelGGL(loop_affine_grad_kernel, dim3((unsigned)((CL + 255) / 256)), dim3(256), 0, st, (const float *)da,
                                   (const float *)m, affine_w, dm, d_affine_w, d_affine_b, first, B, CL);
            }
        }
        MFL_CHECK_HIP(hipGetLastError());
        if (split) {
            rc = mfl::local_backward_apply(xt, img, pos_w, col_w, dm, dx, dpar + 2 * t, B, C, H, W, k, sub, ws.sub_bytes, stream);
        } else {
            rc = mfl_local_backward(xt, img, nullptr, pos_w, col_w, dm, dx, dpar + 2 * t, nullptr, B, C, H, W, k, sub, ws.sub_bytes,
                                    stream);
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(loop_final_kernel, dim3((unsigned)(((size_t)B * L + 255) / 256)), dim3(256), 0, st, unary, dx, dls, d_unary,
                       dpar, dparams, iters, B, C, L);
    return (int)hipGetLastError();
}
