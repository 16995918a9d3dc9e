// Tail pixels of the diagonal / ring gather kernels: the few pixels of a nearly empty last b-block
// (DiagGeom::tail_lanes) computed straight from the index map.  Device code only.
#ifndef MFL_TAIL_H
#define MFL_TAIL_H

#include <hip/hip_runtime.h>

#include "mfl_diag.h"

namespace mfl {

// The few pixels of a nearly empty last b-block (DiagGeom::tail_lanes) that the diagonal kernels
// leave out, computed by extra workgroups at the end of the same launch (they fill the CUs that
// the last round of regular workgroups leaves idle).  One wave per (image, group, chunk of 4 tail pixels); thread = (pixel, one of 16 channel
// shares), so that the lanes of a load touch the two or three rows the group's pixels share instead
// of one row each.  The index map (LocalConvolution.py:59-90 restated in mfl_core.h) is walked
// incrementally over the taps: V index G(c, t+1) = G(c, t) + L, decoded through an LDS table of the
// CK window offsets.  No workgroup barrier after the table: a wave's LDS traffic is ordered.
// MODE 0: fused mean-field update   MODE 1: y = x * window sums
__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

constexpr int kTailPix = 4;      // pixels per wave
constexpr int kTailShares = 16;  // threads per pixel
constexpr int kTailChan = 3;     // channels per thread: the tail path serves up to 48 channels

template <int MODE>
__device__ __forceinline__ void diag_tail_block(const DiagGeom &g, float *tail_lds, int tail_block, int nw, int per_group,
                                                const float *__restrict__ xp, const F4 *__restrict__ ip4,
                                                const float *__restrict__ pos_w, const float *__restrict__ col_w,
                                                const float *__restrict__ unary, const float *__restrict__ aw,
                                                const float *__restrict__ ab, const float *__restrict__ compat,
                                                float *__restrict__ logq, float *__restrict__ qp_out,
                                                float *__restrict__ y_out)
{
    // (the backward scatter, MODE 3, runs every b-block as regular workgroups: its launches have no tail workgroups)
    if constexpr (MODE == 3) return;
    const int chunks = (per_group + kTailPix - 1) / kTailPix;
    const long long nwaves = (long long)g.B * g.NG * chunks;
    const int C = g.C, K = g.K, k = g.k, W = g.W, Wp = g.Wp, L = g.L, CK = g.CK;
    int *tab = reinterpret_cast<int *>(tail_lds);                 // [CK] window offsets
    for (int r = threadIdx.x; r < CK; r += blockDim.x) {
        const int cs = r / K, ts = r - cs * K;
        const int tky = ts / k;
        tab[r] = cs * g.plane + tky * Wp + (ts - tky * k);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, tid = threadIdx.x & 63;
    float *wbase = tail_lds + CK + wave * (kTailPix * (K + 2 * C));
    float *wt = wbase;                          // [pixel][K] tap weights
    float *mm = wbase + kTailPix * K;           // [pixel][C] messages
    float *zz = mm + kTailPix * C;              // [pixel][C] unaries, then logits
    const long long wslot = (long long)tail_block * nw + wave;
    if (wslot >= nwaves) return;
    const int img = (int)(wslot / ((long long)g.NG * chunks));
    const int rem_w = (int)(wslot - (long long)img * g.NG * chunks);
    const int gamma = rem_w / chunks, chunk = rem_w - gamma * chunks;
    const int p = tid / kTailShares, share = tid - p * kTailShares;
    const int pidx = chunk * kTailPix + p;      // pixel of the group's tail pixels
    // (n, lane) of the pixel
    int n = 0, lane = pidx;
    bool valid = pidx < per_group;
    if (valid) while (lane >= diag_tail_lanes(g, n)) lane -= diag_tail_lanes(g, n++);
    const int bb = g.BL * g.NBB + lane - n;
    const long long l64 = (long long)gamma * k + n + (long long)bb * CK;
    valid = valid && l64 < L;
    const int l = valid ? (int)l64 : 0;
    const int py = l / W, px = l - py * W;
    const F4 *ip_b = ip4 + (size_t)img * g.plane;
    const size_t img_off = (size_t)img * C * g.plane;
    const float *in_b = xp + img_off + g.FS;
    const int cpos = (py + g.p0) * Wp + px + g.p0;
    const float pw = *pos_w, cw = *col_w;
    {
        const F4 ctr = ip_b[cpos];
        for (int t = share; t < K; t += kTailShares) {
            const int ky = t / k, kx = t - ky * k;
            const F4 nb = ip_b[(py + ky) * Wp + px + kx];
            const float d0 = ctr.x - nb.x, d1 = ctr.y - nb.y, d2 = ctr.z - nb.z;
            const int dy = ky - g.p0, dx = kx - g.p0;
            wt[p * K + t] = fmaf(fmaf(d2, d2, fmaf(d1, d1, d0 * d0)), cw, (float)(dy * dy + dx * dx) * pw);
        }
    }
    wave_sync();
    // channels share, share + kTailShares, ... (at most kTailChan per thread), walked together so that
    // their loads overlap: V index of (c, t = 0) for this pixel, then + L per tap
    const unsigned qL = (unsigned)(L / CK), rL = (unsigned)(L - (long long)qL * CK);
    unsigned lp[kTailChan], rr[kTailChan];
    float acc[kTailChan], qv[kTailChan], uv[kTailChan], av[kTailChan], bv[kTailChan];
#pragma unroll
    for (int j = 0; j < kTailChan; ++j) {
        const int c = share + j * kTailShares;
        const int cc = c < C ? c : 0;
        const unsigned long long g0 = (unsigned long long)cc * K * (unsigned long long)L + (unsigned long long)l;
        lp[j] = (unsigned)(g0 / (unsigned)CK);
        rr[j] = (unsigned)(g0 - (unsigned long long)lp[j] * (unsigned)CK);
        acc[j] = 0.f;
        // what depends on (pixel, channel) only is requested before the taps
        qv[j] = in_b[(size_t)cc * g.plane + cpos];
        uv[j] = 0.f; av[j] = 1.f; bv[j] = 0.f;
        if constexpr (MODE == 0) uv[j] = unary[((size_t)img * C + cc) * L + l];
        if constexpr (MODE == 0) {
            if (aw) { av[j] = aw[(size_t)cc * L + l]; bv[j] = ab[(size_t)cc * L + l]; }
        }
    }
#pragma unroll 4
    for (int t = 0; t < K; ++t) {
        const float wv = wt[p * K + t];
#pragma unroll
        for (int j = 0; j < kTailChan; ++j) {
            const unsigned ys = lp[j] / (unsigned)W, xs = lp[j] - ys * (unsigned)W;
            acc[j] = fmaf(wv, in_b[tab[rr[j]] + ys * Wp + xs], acc[j]);
            lp[j] += qL;
            rr[j] += rL;
            if (rr[j] >= (unsigned)CK) { rr[j] -= CK; ++lp[j]; }
        }
    }
#pragma unroll
    for (int j = 0; j < kTailChan; ++j) {
        const int c = share + j * kTailShares;
        if (c < C) {
            if constexpr (MODE == 1 || MODE == 2) {
                if (valid) y_out[((size_t)img * C + c) * L + l] = MODE == 1 ? acc[j] * qv[j] : acc[j];
            } else if constexpr (MODE == 0) {
                mm[p * C + c] = fmaf(av[j], acc[j] * qv[j], bv[j]);
                zz[p * C + c] = uv[j];
            }
        }
    }
    if constexpr (MODE == 0) {
        wave_sync();
        // every thread of the pixel reduces over all channels; it finishes its own channels
        float mu = -INFINITY;
        for (int c2 = 0; c2 < C; ++c2) mu = fmaxf(mu, zz[p * C + c2]);
        float su = 0.f;
        for (int c2 = 0; c2 < C; ++c2) su += __expf(zz[p * C + c2] - mu);
        const float lse_u = mu + __logf(su);
        float zmine[kTailChan];
#pragma unroll
        for (int j = 0; j < kTailChan; ++j) {
            const int c = share + j * kTailShares;
            zmine[j] = 0.f;
            if (c < C) {
                float pv = mm[p * C + c];
                if (compat) {
                    pv = 0.f;
                    for (int c2 = 0; c2 < C; ++c2) pv = fmaf(compat[c * C + c2], mm[p * C + c2], pv);
                }
                zmine[j] = (zz[p * C + c] - lse_u) - pv;
            }
        }
        wave_sync();
#pragma unroll
        for (int j = 0; j < kTailChan; ++j) {
            const int c = share + j * kTailShares;
            if (c < C) zz[p * C + c] = zmine[j];
        }
        wave_sync();
        float mv = -INFINITY;
        for (int c2 = 0; c2 < C; ++c2) mv = fmaxf(mv, zz[p * C + c2]);
        float sv = 0.f;
        for (int c2 = 0; c2 < C; ++c2) sv += __expf(zz[p * C + c2] - mv);
        const float lse_v = mv + __logf(sv);
#pragma unroll
        for (int j = 0; j < kTailChan; ++j) {
            const int c = share + j * kTailShares;
            const float lq = zmine[j] - lse_v;
            if (valid && c < C) {
                if (logq) logq[((size_t)img * C + c) * L + l] = lq;
                qp_out[img_off + g.FS + (size_t)c * g.plane + cpos] = __expf(lq);
            }
        }
    }
}

}  // namespace mfl

#endif  // MFL_TAIL_H
