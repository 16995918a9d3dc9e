// Gather kernels of the local-window message-passing library (gfx950), one
// translation unit per (channel-chunk width, kernel kind): compile with -DMFL_CC=<8|19|21|32>
// -DMFL_KIND=<1 forward | 2 backward | 3 fused mean-field step>.
// Thread mapping, index map and incremental decode: mfl_core.h.  Data layout and
// the C ABI: mfl_api.hip, include/mfl.h.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfl.h"
#include "mfl_gather.h"

#if !defined(MFL_CC) || !defined(MFL_KIND)
#error "compile with -DMFL_CC=<channel chunk width> -DMFL_KIND=<1 forward | 2 backward | 3 fused step>"
#endif

using namespace mfl;

namespace {


template <bool TAB_LDS>
__device__ __forceinline__ const int *stage_table(const int *__restrict__ tab_g, int *lds, int CK)
{
    if constexpr (TAB_LDS) {
        for (int r = threadIdx.x; r < CK; r += kThreads) lds[r] = tab_g[r];
        __syncthreads();
        return lds;
    } else {
        return tab_g;
    }
}

#if MFL_KIND != 3
// tap-split kernels (large windows on small images): threads per pixel, pixels per workgroup
constexpr int kTapSlices = 16, kTapPixels = kThreads / kTapSlices;
#endif

#if MFL_KIND == 1
// y[b][c][l] = x[b][c][l] * sum_t w[b][l][t] * S[b][c][l][t]
// grid: (blocks_per_image, channel chunks, B)
template <int CC, int M, bool FROM_CD, bool TAB_LDS, bool EXACT>
__global__ void __launch_bounds__(kThreads)
local_forward_kernel(Geom g, const float *__restrict__ xp, const F4 *__restrict__ ip4, const float *__restrict__ cd,
                     const int *__restrict__ tab_g, const float *__restrict__ pos_w, const float *__restrict__ col_w,
                     float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) int lds_tab[];
    const int *tab = stage_table<TAB_LDS>(tab_g, lds_tab, g.CK);

    const int b = blockIdx.z, c0 = blockIdx.y * CC;
    const int nch = min(CC, g.C - c0);
    Columns<M> col;
    make_columns<M>(col, g, blockIdx.x, threadIdx.x);

    const float *xp_b = xp + (size_t)b * g.C * g.plane;
    const F4 *ip_b = FROM_CD ? nullptr : ip4 + (size_t)b * g.plane;
    const float *cd_b = FROM_CD ? cd + (size_t)b * g.L * g.K : nullptr;

    float acc[CC][M];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int n = 0; n < M; ++n) acc[cc][n] = 0.f;

    NoSink sink;
    if constexpr (TAB_LDS)
        window_sums<CC, M, FROM_CD, EXACT>(acc, g, xp_b, ip_b, cd_b, lds_tab, *pos_w, *col_w, c0, nch, col, nullptr, sink);
    else
        window_sums<CC, M, FROM_CD, EXACT>(acc, g, xp_b, ip_b, cd_b, tab, *pos_w, *col_w, c0, nch, col, nullptr, sink);

    const int ctr_off = g.p0 * g.Wp + g.p0;
    float *y_b = y + (size_t)b * g.C * g.L;
#pragma unroll
    for (int n = 0; n < M; ++n) {
        if (col.valid[n]) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nch) {
                    const int c = c0 + cc;
                    y_b[(size_t)c * g.L + col.lcl[n]] = acc[cc][n] * xp_b[(size_t)c * g.plane + col.ibase[n] + ctr_off];
                }
        }
    }
}

// Large windows on small images (the reference model's own 50 x 50 window on a 64 x 64 image, AdvancedModel.py:135-137):
// with one pixel per thread a thread walks K dependent taps (a table entry, then the element it addresses), which is
// bound by that latency, and the few pixels leave most CUs idle.  Here the taps of a pixel are split over kTapSlices
// threads (kTapPixels pixels per workgroup); the partial sums meet in LDS and are added in a fixed order.
// Same index walk as window_sums (mfl_core.h) with one pixel per thread, started at the slice's first tap.
template <int CC>
__global__ void __launch_bounds__(kThreads)
local_forward_taps_kernel(Geom g, const float *__restrict__ xp, const F4 *__restrict__ ip4, const int *__restrict__ tab,
                          const float *__restrict__ pos_w, const float *__restrict__ col_w, float *__restrict__ y)
{
    __shared__ float red[kTapSlices][CC][kTapPixels + 1];
    const int tid = threadIdx.x, pxi = tid % kTapPixels, slice = tid / kTapPixels;
    const int b = blockIdx.z, c0 = blockIdx.y * CC;
    const int nch = min(CC, g.C - c0);
    const int l_raw = blockIdx.x * kTapPixels + pxi;
    const int l = l_raw < g.L ? l_raw : g.L - 1;
    const int W = g.W, Wp = g.Wp, gap = Wp - W, CK = g.CK;
    const int py = l / W, px = l - py * W;
    const int ibase = py * Wp + px;
    const float *xp_b = xp + (size_t)b * g.C * g.plane;
    const F4 *ip_b = ip4 + (size_t)b * g.plane;
    const float pw = *pos_w, cw = *col_w;
    const int t0 = (int)((long long)g.K * slice / kTapSlices), t1 = (int)((long long)g.K * (slice + 1) / kTapSlices);
    // decode state of (channel c0, tap t0): table index rs, source column xs, padded offset ls = y' * Wp + x'
    const long long f0 = ((long long)c0 * g.K + t0) * g.L + l;
    const long long lp0 = f0 / CK;
    int rs = (int)(f0 - lp0 * CK);
    int xs = (int)(lp0 % W);
    int ls = (int)(lp0 / W) * Wp + xs;
    const int dq_lin = g.dqy * Wp + g.dqx, dK_lin = g.dKy * Wp + g.dKx;
    float acc[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) acc[cc] = 0.f;
    const F4 ctr = ip_b[ibase + g.p0 * Wp + g.p0];
    int ky = t0 / g.k, kx = t0 - ky * g.k;
    for (int t = t0; t < t1; ++t) {
        const F4 nb = ip_b[ibase + ky * Wp + kx];
        const float d0 = ctr.x - nb.x, d1 = ctr.y - nb.y, d2 = ctr.z - nb.z;
        const int dy = ky - g.p0, dx = kx - g.p0;
        const float w = fmaf(d0 * d0 + d1 * d1 + d2 * d2, cw, (float)(dy * dy + dx * dx) * pw);
        int r = rs, x = xs, lin = ls;
        int a[CC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            a[cc] = lin + tab[r];
            if (cc >= nch) a[cc] = a[0];          // channels past the chunk: a valid dummy
            r += g.rhoK;
            const int carry = r >= CK;
            r -= carry ? CK : 0;
            x += g.dKx + carry;
            const int wrap = x >= W;
            x -= wrap ? W : 0;
            lin += dK_lin + carry + (wrap ? gap : 0);
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) acc[cc] = fmaf(w, xp_b[a[cc]], acc[cc]);
        // decode step to the next tap
        rs += g.rho;
        const int carry = rs >= CK;
        rs -= carry ? CK : 0;
        xs += g.dqx + carry;
        const int wrap = xs >= W;
        xs -= wrap ? W : 0;
        ls += dq_lin + carry + (wrap ? gap : 0);
        if (++kx == g.k) { kx = 0; ++ky; }
    }
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) red[slice][cc][pxi] = acc[cc];
    __syncthreads();
    if (l_raw < g.L) {
        const int ctr_off = g.p0 * Wp + g.p0;
        for (int cc = slice; cc < nch; cc += kTapSlices) {
            float sum = 0.f;
#pragma unroll
            for (int i = 0; i < kTapSlices; ++i) sum += red[i][cc][pxi];
            const int c = c0 + cc;
            y[((size_t)b * g.C + c) * g.L + l] = sum * xp_b[(size_t)c * g.plane + ibase + ctr_off];
        }
    }
}

template <int CC>
int launch_fwd_taps(const GatherArgs &a, float *y, hipStream_t st)
{
    const Geom &g = a.g;
    const dim3 grid((g.L + kTapPixels - 1) / kTapPixels, (g.C + CC - 1) / CC, g.B);
    hipLaunchKernelGGL((local_forward_taps_kernel<CC>), grid, dim3(kThreads), 0, st, g, a.xp, a.ip4, a.tab, a.pos_w, a.col_w, y);
    return (int)hipGetLastError();
}

#endif  // MFL_KIND == 1

#if MFL_KIND == 3
// One fused mean-field update (all C <= CC channels of a pixel live in one thread):
//   msg = Q_t * window_sum(Q_t);  msg = aw*msg + ab;  P = compat . msg;
//   logQ = LogSoftmax(log_softmax(U) - P);  Q_{t+1} = exp(logQ) -> interior of qp_out
// grid: (blocks_per_image, 1, B).  LDS: [offset table][compat matrix].
template <int CC, int M, bool TAB_LDS, bool EXACT>
__global__ void __launch_bounds__(kThreads)
meanfield_step_kernel(Geom g, const float *__restrict__ qp_in, const F4 *__restrict__ ip4,
                      const int *__restrict__ tab_g, const float *__restrict__ pos_w, const float *__restrict__ col_w,
                      const float *__restrict__ unary, const float *__restrict__ aw, const float *__restrict__ ab,
                      const float *__restrict__ compat, float *__restrict__ logq, float *__restrict__ qp_out)
{
    extern __shared__ __attribute__((aligned(16))) int lds_tab[];
    float *lds_compat = reinterpret_cast<float *>(lds_tab + (TAB_LDS ? ((g.CK + 3) & ~3) : 0));
    if (compat)
        for (int i = threadIdx.x; i < g.C * g.C; i += kThreads) lds_compat[i] = compat[i];
    const int *tab = stage_table<TAB_LDS>(tab_g, lds_tab, g.CK);
    if constexpr (!TAB_LDS) __syncthreads();

    const int b = blockIdx.z, C = g.C;
    Columns<M> col;
    make_columns<M>(col, g, blockIdx.x, threadIdx.x);

    const float *qp_b = qp_in + (size_t)b * C * g.plane;
    const F4 *ip_b = ip4 + (size_t)b * g.plane;

    float acc[CC][M];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int n = 0; n < M; ++n) acc[cc][n] = 0.f;

    NoSink sink;
    if constexpr (TAB_LDS)
        window_sums<CC, M, false, EXACT>(acc, g, qp_b, ip_b, nullptr, lds_tab, *pos_w, *col_w, 0, C, col, nullptr, sink);
    else
        window_sums<CC, M, false, EXACT>(acc, g, qp_b, ip_b, nullptr, tab, *pos_w, *col_w, 0, C, col, nullptr, sink);

    const int ctr_off = g.p0 * g.Wp + g.p0;
    const float *u_b = unary + (size_t)b * C * g.L;
    float *qo_b = qp_out + (size_t)b * C * g.plane;
#pragma unroll
    for (int n = 0; n < M; ++n) {
        if (!col.valid[n]) continue;
        const int l = col.lcl[n], ci = col.ibase[n] + ctr_off;
        float m[CC], u[CC];
        float mu = -INFINITY;
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < C) {
                float v = acc[c][n] * qp_b[(size_t)c * g.plane + ci];
                if (aw) v = fmaf(aw[(size_t)c * g.L + l], v, ab[(size_t)c * g.L + l]);
                m[c] = v;
                u[c] = u_b[(size_t)c * g.L + l];
                mu = fmaxf(mu, u[c]);
            }
        float su = 0.f;
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < C) su += expf(u[c] - mu);
        const float lse_u = mu + logf(su);
        float mv = -INFINITY;
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < C) {
                float p;
                if (compat) {
                    p = 0.f;
#pragma unroll
                    for (int c2 = 0; c2 < CC; ++c2)
                        if (c2 < C) p = fmaf(lds_compat[c * C + c2], m[c2], p);
                } else {
                    p = m[c];
                }
                u[c] = (u[c] - lse_u) - p;
                mv = fmaxf(mv, u[c]);
            }
        float sv = 0.f;
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < C) sv += expf(u[c] - mv);
        const float lse_v = mv + logf(sv);
#pragma unroll
        for (int c = 0; c < CC; ++c)
            if (c < C) {
                const float lq = u[c] - lse_v;
                if (logq) logq[((size_t)b * C + c) * g.L + l] = lq;
                qo_b[(size_t)c * g.plane + ci] = expf(lq);
            }
    }
}

#endif  // MFL_KIND == 3

#if MFL_KIND == 2
// Backward of local_forward_kernel.  Mirrors the forward loop: the same gather gives
// S, from which  out (for dx = dy*out), u_t = sum_c D*S (for the two parameter
// gradients and dcd) follow, and the scatter half of dx is added with float atomics
// into zero-initialised padded planes dxp.  D = dy * x.
//   dxd       [B][C][H][W]   <- dy * out        (plain stores, one owner per element)
//   dxp       [B][C][Hp][Wp] += D * w           (atomics)
//   partials  [blocks][2]    <- per-block sums of {d pos_w, d col_w}
template <int M>
struct BackwardSink {
    static constexpr bool enabled = true;
    float *dxp_b;
    const float (*D)[M];
    const bool *valid;
    int W, gap;
    float tpw, tcw;
    float *dcd_b;      // optional [L][K]
    const int *lcl;
    int K;
    float cw;
    bool dcd_atomic;

    __device__ __forceinline__ void scatter(int a, int x, int cc, const float (&w)[M])
    {
#pragma unroll
        for (int n = 0; n < M; ++n)
            if (valid[n]) atomicAdd(dxp_b + a + n + ((x + n >= W) ? gap : 0), w[n] * D[cc][n]);
    }
    __device__ __forceinline__ void tap(int t, float pd, const float (&cdv)[M], const float (&u)[M])
    {
#pragma unroll
        for (int n = 0; n < M; ++n)
            if (valid[n]) {
                tpw = fmaf(pd, u[n], tpw);
                tcw = fmaf(cdv[n], u[n], tcw);
                if (dcd_b) {
                    float *p = dcd_b + (size_t)lcl[n] * K + t;
                    if (dcd_atomic) atomicAdd(p, cw * u[n]);
                    else *p = cw * u[n];
                }
            }
    }
};

template <int CC, int M, bool FROM_CD, bool TAB_LDS, bool EXACT>
__global__ void __launch_bounds__(kThreads)
local_backward_kernel(Geom g, const float *__restrict__ xp, const F4 *__restrict__ ip4, const float *__restrict__ cd,
                      const int *__restrict__ tab_g, const float *__restrict__ pos_w, const float *__restrict__ col_w,
                      const float *__restrict__ dy, float *__restrict__ dxd, float *__restrict__ dxp,
                      float *__restrict__ partials, float *__restrict__ dcd)
{
    extern __shared__ __attribute__((aligned(16))) int lds_tab[];
    __shared__ float red[2][kThreads / 64];
    const int *tab = stage_table<TAB_LDS>(tab_g, lds_tab, g.CK);

    const int b = blockIdx.z, c0 = blockIdx.y * CC;
    const int nch = min(CC, g.C - c0);
    Columns<M> col;
    make_columns<M>(col, g, blockIdx.x, threadIdx.x);

    const float *xp_b = xp + (size_t)b * g.C * g.plane;
    const F4 *ip_b = FROM_CD ? nullptr : ip4 + (size_t)b * g.plane;
    const float *cd_b = FROM_CD ? cd + (size_t)b * g.L * g.K : nullptr;
    const float *dy_b = dy + (size_t)b * g.C * g.L;
    const int ctr_off = g.p0 * g.Wp + g.p0;

    float acc[CC][M], D[CC][M], dyv[CC][M];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int n = 0; n < M; ++n) {
            acc[cc][n] = 0.f;
            float d = 0.f, xv = 0.f;
            if (cc < nch && col.valid[n]) {
                d = dy_b[(size_t)(c0 + cc) * g.L + col.lcl[n]];
                xv = xp_b[(size_t)(c0 + cc) * g.plane + col.ibase[n] + ctr_off];
            }
            dyv[cc][n] = d;
            D[cc][n] = d * xv;
        }

    BackwardSink<M> sink;
    sink.dxp_b = dxp + (size_t)b * g.C * g.plane;
    sink.D = D;
    sink.valid = col.valid;
    sink.W = g.W;
    sink.gap = g.Wp - g.W;
    sink.tpw = 0.f;
    sink.tcw = 0.f;
    sink.dcd_b = dcd ? dcd + (size_t)b * g.L * g.K : nullptr;
    sink.lcl = col.lcl;
    sink.K = g.K;
    sink.cw = *col_w;
    sink.dcd_atomic = gridDim.y > 1;

    if constexpr (TAB_LDS)
        window_sums<CC, M, FROM_CD, EXACT>(acc, g, xp_b, ip_b, cd_b, lds_tab, *pos_w, *col_w, c0, nch, col, D, sink);
    else
        window_sums<CC, M, FROM_CD, EXACT>(acc, g, xp_b, ip_b, cd_b, tab, *pos_w, *col_w, c0, nch, col, D, sink);

    float *dxd_b = dxd + (size_t)b * g.C * g.L;
#pragma unroll
    for (int n = 0; n < M; ++n)
        if (col.valid[n]) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nch) dxd_b[(size_t)(c0 + cc) * g.L + col.lcl[n]] = dyv[cc][n] * acc[cc][n];
        }

    // block reduction of the two parameter-gradient partial sums (fixed order: deterministic)
    float tpw = sink.tpw, tcw = sink.tcw;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        tpw += __shfl_down(tpw, off, 64);
        tcw += __shfl_down(tcw, off, 64);
    }
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (lane == 0) { red[0][wave] = tpw; red[1][wave] = tcw; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, c = 0.f;
        for (int i = 0; i < kThreads / 64; ++i) { a += red[0][i]; c += red[1][i]; }
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        partials[2 * blk] = a;
        partials[2 * blk + 1] = c;
    }
}

// Backward for large windows on small images (the reference model's own 50 x 50 window on 64 x 64, AdvancedModel.py:135-137):
// the tap-split walk of local_forward_taps_kernel, transposed.  A pixel's taps are split over kTapSlices threads; each
// keeps, per channel, the plain window sums with the weights pd[t] alone (P) and cd[l][t] alone (Q), which meet in LDS
// and are added in a fixed order, and adds what its pixel sends to every element it reads, w[t] * dy * x, to the
// gradient planes in 64-bit fixed point (integer atomics: the sums do not depend on the order).  dx and both parameter
// gradients follow from P, Q and those planes (diag_backward_finish_kernel in mfl_api.hip):
//   dx = dy * (pos_w * P + col_w * Q) + planes,  d pos_w = sum dy * x * P,  d col_w = sum dy * x * Q   (LocalConvolution.py:84-90)
template <int CC>
__global__ void __launch_bounds__(kThreads)
local_backward_taps_kernel(Geom g, const float *__restrict__ xp, const F4 *__restrict__ ip4, const int *__restrict__ tab,
                           const float *__restrict__ pos_w, const float *__restrict__ col_w, const float *__restrict__ dy,
                           const float *__restrict__ scale, float *__restrict__ P, float *__restrict__ Q,
                           unsigned long long *__restrict__ dxp)
{
    __shared__ float red[kTapSlices][CC][kTapPixels + 1];
    const int tid = threadIdx.x, pxi = tid % kTapPixels, slice = tid / kTapPixels;
    const int b = blockIdx.z, c0 = blockIdx.y * CC;
    const int nch = min(CC, g.C - c0);
    const int l_raw = blockIdx.x * kTapPixels + pxi;
    const bool live = l_raw < g.L;
    const int l = live ? l_raw : g.L - 1;
    const int W = g.W, Wp = g.Wp, gap = Wp - W, CK = g.CK;
    const int py = l / W, px = l - py * W;
    const int ibase = py * Wp + px;
    const float *xp_b = xp + (size_t)b * g.C * g.plane;
    const F4 *ip_b = ip4 + (size_t)b * g.plane;
    unsigned long long *dxp_b = dxp + (size_t)b * g.C * g.plane;
    const float pw = *pos_w, cw = *col_w, fscale = *scale;
    const int t0 = (int)((long long)g.K * slice / kTapSlices), t1 = (int)((long long)g.K * (slice + 1) / kTapSlices);
    const long long f0 = ((long long)c0 * g.K + t0) * g.L + l;
    const long long lp0 = f0 / CK;
    int rs = (int)(f0 - lp0 * CK);
    int xs = (int)(lp0 % W);
    int ls = (int)(lp0 / W) * Wp + xs;
    const int dq_lin = g.dqy * Wp + g.dqx, dK_lin = g.dKy * Wp + g.dKx;
    const int ctr_off = g.p0 * Wp + g.p0;
    float pacc[CC], qacc[CC], D[CC];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc) {
        pacc[cc] = 0.f;
        qacc[cc] = 0.f;
        D[cc] = (live && cc < nch) ? dy[((size_t)b * g.C + c0 + cc) * g.L + l] * xp_b[(size_t)(c0 + cc) * g.plane + ibase + ctr_off] : 0.f;
    }
    const F4 ctr = ip_b[ibase + ctr_off];
    int ky = t0 / g.k, kx = t0 - ky * g.k;
    for (int t = t0; t < t1; ++t) {
        const F4 nb = ip_b[ibase + ky * Wp + kx];
        const float d0 = ctr.x - nb.x, d1 = ctr.y - nb.y, d2 = ctr.z - nb.z;
        const int dyy = ky - g.p0, dxx = kx - g.p0;
        const float cd = d0 * d0 + d1 * d1 + d2 * d2, pd = (float)(dyy * dyy + dxx * dxx);
        const float w = fmaf(cd, cw, pd * pw);
        int r = rs, x = xs, lin = ls;
        int a[CC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            a[cc] = lin + tab[r];
            if (cc >= nch) a[cc] = a[0];          // channels past the chunk: a valid dummy (their D is 0)
            r += g.rhoK;
            const int carry = r >= CK;
            r -= carry ? CK : 0;
            x += g.dKx + carry;
            const int wrap = x >= W;
            x -= wrap ? W : 0;
            lin += dK_lin + carry + (wrap ? gap : 0);
        }
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
            const float v = xp_b[a[cc]];
            pacc[cc] = fmaf(pd, v, pacc[cc]);
            qacc[cc] = fmaf(cd, v, qacc[cc]);
            const float s = w * D[cc];
            if (s != 0.f) fixed_point_add(dxp_b + a[cc], s, fscale);
        }
        rs += g.rho;
        const int carry = rs >= CK;
        rs -= carry ? CK : 0;
        xs += g.dqx + carry;
        const int wrap = xs >= W;
        xs -= wrap ? W : 0;
        ls += dq_lin + carry + (wrap ? gap : 0);
        if (++kx == g.k) { kx = 0; ++ky; }
    }
    // the slices' partial sums meet in LDS and are added in slice order: P first, then Q through the same buffer
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) red[slice][cc][pxi] = pass ? qacc[cc] : pacc[cc];
        __syncthreads();
        if (live) {
            for (int cc = slice; cc < nch; cc += kTapSlices) {
                float sum = 0.f;
#pragma unroll
                for (int i = 0; i < kTapSlices; ++i) sum += red[i][cc][pxi];
                (pass ? Q : P)[((size_t)b * g.C + c0 + cc) * g.L + l] = sum;
            }
        }
    }
}

template <int CC>
int launch_bwd_taps(const GatherArgs &a, const TapsBackwardArgs &t, hipStream_t st)
{
    const Geom &g = a.g;
    const dim3 grid((g.L + kTapPixels - 1) / kTapPixels, (g.C + CC - 1) / CC, g.B);
    hipLaunchKernelGGL((local_backward_taps_kernel<CC>), grid, dim3(kThreads), 0, st, g, a.xp, a.ip4, a.tab, a.pos_w, a.col_w, t.dy,
                       t.scale, t.P, t.Q, t.dxp);
    return (int)hipGetLastError();
}

#endif  // MFL_KIND == 2


#if MFL_KIND == 1
template <int CC, int M, bool FROM_CD, bool TAB_LDS>
int launch_fwd(const GatherArgs &a, float *y, hipStream_t st)
{
    const Geom &g = a.g;
    if (g.M != M) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid(blocks_per_image(g), (g.C + CC - 1) / CC, g.B);
    const size_t lds = TAB_LDS ? (size_t)g.CK * 4 : 0;
    if (g.C % CC == 0)   // every chunk is full: no per-channel range checks in the kernel
        hipLaunchKernelGGL((local_forward_kernel<CC, M, FROM_CD, TAB_LDS, true>), grid, dim3(kThreads), lds, st, g,
                           a.xp, a.ip4, a.cd, a.tab, a.pos_w, a.col_w, y);
    else
        hipLaunchKernelGGL((local_forward_kernel<CC, M, FROM_CD, TAB_LDS, false>), grid, dim3(kThreads), lds, st, g,
                           a.xp, a.ip4, a.cd, a.tab, a.pos_w, a.col_w, y);
    return (int)hipGetLastError();
}

#endif  // MFL_KIND == 1

#if MFL_KIND == 2
template <int CC, int M, bool FROM_CD, bool TAB_LDS>
int launch_bwd(const GatherArgs &a, const BackwardArgs &b, hipStream_t st)
{
    const Geom &g = a.g;
    if (g.M != M) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid(blocks_per_image(g), (g.C + CC - 1) / CC, g.B);
    const size_t lds = TAB_LDS ? (size_t)g.CK * 4 : 0;
    if (g.C % CC == 0)
        hipLaunchKernelGGL((local_backward_kernel<CC, M, FROM_CD, TAB_LDS, true>), grid, dim3(kThreads), lds, st, g,
                           a.xp, a.ip4, a.cd, a.tab, a.pos_w, a.col_w, b.dy, b.dxd, b.dxp, b.partials, b.dcd);
    else
        hipLaunchKernelGGL((local_backward_kernel<CC, M, FROM_CD, TAB_LDS, false>), grid, dim3(kThreads), lds, st, g,
                           a.xp, a.ip4, a.cd, a.tab, a.pos_w, a.col_w, b.dy, b.dxd, b.dxp, b.partials, b.dcd);
    return (int)hipGetLastError();
}

#endif  // MFL_KIND == 2

#if MFL_KIND == 3
template <int CC, int M, bool TAB_LDS>
int launch_step(const GatherArgs &a, const StepArgs &s, hipStream_t st)
{
    const Geom &g = a.g;
    if (g.M != M || g.C > CC) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid(blocks_per_image(g), 1, g.B);
    const size_t lds = (TAB_LDS ? (size_t)((g.CK + 3) & ~3) * 4 : 0) + (size_t)g.C * g.C * 4;
    if (g.C == CC)
        hipLaunchKernelGGL((meanfield_step_kernel<CC, M, TAB_LDS, true>), grid, dim3(kThreads), lds, st, g, a.xp,
                           a.ip4, a.tab, a.pos_w, a.col_w, s.unary, s.aw, s.ab, s.compat, s.logq, s.qp_out);
#ifndef MFL_DEV_FAST
    else
        hipLaunchKernelGGL((meanfield_step_kernel<CC, M, TAB_LDS, false>), grid, dim3(kThreads), lds, st, g, a.xp,
                           a.ip4, a.tab, a.pos_w, a.col_w, s.unary, s.aw, s.ab, s.compat, s.logq, s.qp_out);
#endif
    return (int)hipGetLastError();
}

#endif  // MFL_KIND == 3

// Pixels per thread the image-weight, LDS-table variants are built for at this width
// (accumulator tile CC*M kept within the register budget).
[[maybe_unused]] constexpr int kMaxM = (MFL_CC * 4 <= 96) ? 4 : 2;

}  // namespace

namespace mfl {

#define MFL_CAT2(a, b) a##b
#define MFL_CAT(a, b) MFL_CAT2(a, b)

#if MFL_KIND == 1
int MFL_CAT(launch_local_forward_cc, MFL_CC)(const GatherArgs &a, float *y, hipStream_t st)
{
    constexpr int CC = MFL_CC;
    if (a.from_cd)   // caller-supplied tensor: compatibility path, one pixel per thread
        return a.tab_lds ? launch_fwd<CC, 1, true, true>(a, y, st) : launch_fwd<CC, 1, true, false>(a, y, st);
    if (taps_kernel_applies(a.g)) return launch_fwd_taps<CC>(a, y, st);
    if (!a.tab_lds) return launch_fwd<CC, 1, false, false>(a, y, st);
    switch (a.g.M) {
    case 1: return launch_fwd<CC, 1, false, true>(a, y, st);
    case 2: return launch_fwd<CC, 2, false, true>(a, y, st);
    case 4: if constexpr (kMaxM >= 4) return launch_fwd<CC, 4, false, true>(a, y, st);
    }
    return MFL_ERR_INVALID_ARGUMENT;
}

#endif  // MFL_KIND == 1

#if MFL_KIND == 2
int MFL_CAT(launch_local_backward_cc, MFL_CC)(const GatherArgs &a, const BackwardArgs &b, hipStream_t st)
{
    constexpr int CC = MFL_CC;
    if (a.from_cd)
        return a.tab_lds ? launch_bwd<CC, 1, true, true>(a, b, st) : launch_bwd<CC, 1, true, false>(a, b, st);
    if (!a.tab_lds) return launch_bwd<CC, 1, false, false>(a, b, st);
    switch (a.g.M) {
    case 1: return launch_bwd<CC, 1, false, true>(a, b, st);
    case 2: return launch_bwd<CC, 2, false, true>(a, b, st);
    }
    return MFL_ERR_INVALID_ARGUMENT;
}

int MFL_CAT(launch_local_backward_taps_cc, MFL_CC)(const GatherArgs &a, const TapsBackwardArgs &t, hipStream_t st)
{
    return launch_bwd_taps<MFL_CC>(a, t, st);
}

#endif  // MFL_KIND == 2

#if MFL_KIND == 3
int MFL_CAT(launch_meanfield_step_cc, MFL_CC)(const GatherArgs &a, const StepArgs &s, hipStream_t st)
{
    constexpr int CC = MFL_CC;
#ifdef MFL_DEV_FAST   // kernel experiments: only the M = MFL_DEV_FAST, LDS-table, exact-width variant is built
    if (a.tab_lds && a.g.M == MFL_DEV_FAST && a.g.C == CC) return launch_step<CC, MFL_DEV_FAST, true>(a, s, st);
    return MFL_ERR_INVALID_ARGUMENT;
#else
    if (!a.tab_lds) return launch_step<CC, 1, false>(a, s, st);
    switch (a.g.M) {
    case 1: return launch_step<CC, 1, true>(a, s, st);
    case 2: return launch_step<CC, 2, true>(a, s, st);
    case 4: if constexpr (kMaxM >= 4) return launch_step<CC, 4, true>(a, s, st);
    }
    return MFL_ERR_INVALID_ARGUMENT;
#endif
}

#endif  // MFL_KIND == 3

}  // namespace mfl
