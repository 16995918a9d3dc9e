// Geometry and the per-thread gather core of the local-window message-passing
// kernels.  Shared by the HIP kernels (device) and by the host emulation used in
// the CPU test-suite, so it is written in plain C++ with a few macros.
//
// Arithmetic restated (not code) from Conv2dLocal.forward, LocalConvolution.py:59-90.
//
// ---- the index map ---------------------------------------------------------
// With K = k*k taps, L = H*W pixels and CK = C*K, the reference's
// Unfold -> permute(0,2,1) -> reshape(B,C,K,-1) makes output (c, l), tap t read
//     V[g],  g = (c*K + t)*L + l,
// where V is the im2col array in [pixel l'][channel c'][tap t'] order:
//     V[l'*CK + c'*K + t'] = pad(x)[c'][l'/W + t'/k][l'%W + t'%k].
// Row j = c*K + t of the [CK][L] view of V therefore holds, for consecutive
// output pixels l, consecutive (c', t') entries of one or two source pixels l'.
//
// ---- thread mapping ---------------------------------------------------------
// A thread owns M output pixels l_n = l_0 + n*CK (n < M) and all channels of a
// channel chunk.  For any row j the M values it needs are V[g_0 + n*CK]: the
// SAME (c', t') of M horizontally consecutive source pixels l', l'+1, ...,
// i.e. M consecutive floats of one padded row (unless the run crosses the end
// of an image row).  One address decode therefore serves M gathers, and the
// gathers are one M-wide load.  A "unit" is a (tile i, t_0) pair, t_0 < CK:
//     l_0 = i*M*CK + t_0;  units are laid out over lanes, waves and blocks by make_columns.
//
// ---- incremental decode ------------------------------------------------------
// Per thread the decode state of row j is (r, y', x') with g_0 = l'*CK + r,
// l' = y'*W + x'.  Moving to the next channel (j += K) or the next tap
// (j += 1) adds a constant to g_0, so the state advances with adds and
// carries only; host-computed constants:
//     L   = q *CK + rho,   q  = dqy*W + dqx       (tap step)
//     K*L = qK*CK + rhoK,  qK = dKy*W + dKx       (channel step)
// The address of the run is  (y'*Wp + x') + tab[r]  in the padded plane set of
// the image, tab[r] = c'*plane + (t'/k)*Wp + t'%k  (a CK-entry table); the
// thread carries y'*Wp + x' as one linear offset plus x' for the row wrap.
#ifndef MFL_CORE_H
#define MFL_CORE_H

#include <stddef.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MFL_HD __host__ __device__ __forceinline__
#define MFL_D __device__ __forceinline__
#else
#include <math.h>
#define MFL_HD inline
#define MFL_D inline
#endif

namespace mfl {

struct alignas(16) F4 {
    float x, y, z, w;
};

// 4-byte aligned vector views: gfx950 global loads only need dword alignment.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
typedef float f2u __attribute__((ext_vector_type(2), aligned(4)));

struct Geom {
    int B, C, H, W, k, K, p0;
    int L, CK;
    int Hp, Wp, plane;  // zero-padded plane: Hp = H+k-1 rows of Wp = W+k-1 floats
    int M, ntile, units;
    int q, rho, dqy, dqx;
    int qK, rhoK, dKy, dKx;
};

// Floats of slack required after the last padded plane: tail threads and
// out-of-range columns of the last tile read (never use) up to M pixels past
// the last source pixel.
MFL_HD size_t padded_slack(const Geom &g) { return (size_t)3 * g.Wp + 64; }

// Pick M (1, 2 or 4 pixels per thread) for a problem: the largest M whose
// tiles are at least 85 % full.
inline int choose_m(long long L, long long CK, int max_m)
{
    for (int m = max_m; m > 1; m >>= 1) {
        long long tile = (long long)m * CK;
        long long nt = (L + tile - 1) / tile;
        if ((double)L / (double)(nt * tile) >= 0.85) return m;
    }
    return 1;
}

// Returns 0 on success, -1 for invalid sizes, -3 when 32-bit indexing would overflow.
inline int make_geom(Geom &g, int B, int C, int H, int W, int k, int M)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || M <= 0) return -1;
    const long long K = (long long)k * k, L = (long long)H * W, CK = (long long)C * K;
    const long long Hp = (long long)H + k - 1, Wp = (long long)W + k - 1, plane = Hp * Wp;
    if (L >= (1LL << 30) || CK >= (1LL << 30) || C * plane + 3 * Wp + 64 >= (1LL << 29)) return -3;
    g.B = B; g.C = C; g.H = H; g.W = W; g.k = k; g.K = (int)K; g.p0 = k / 2;
    g.L = (int)L; g.CK = (int)CK; g.Hp = (int)Hp; g.Wp = (int)Wp; g.plane = (int)plane;
    g.M = M;
    const long long tile = (long long)M * CK;
    g.ntile = (int)((L + tile - 1) / tile);
    if ((long long)g.ntile * CK >= (1LL << 31)) return -3;
    g.units = (int)((long long)g.ntile * CK);
    g.q = (int)(L / CK); g.rho = (int)(L % CK);
    g.dqy = g.q / W; g.dqx = g.q % W;
    const long long KL = K * L;
    g.qK = (int)(KL / CK); g.rhoK = (int)(KL % CK);
    g.dKy = g.qK / W; g.dKx = g.qK % W;
    // initial decode uses c0*rhoK + l0 in 32 bits
    if ((long long)C * g.rhoK + L + CK >= (1LL << 31)) return -3;
    return 0;
}

MFL_HD int table_entry(const Geom &g, int r)
{
    const int cp = r / g.K, tp = r - cp * g.K;
    const int ky = tp / g.k, kx = tp - ky * g.k;
    return cp * g.plane + ky * g.Wp + kx;
}

// Wide load of the M consecutive floats of a run (no row crossing).
template <int M>
MFL_D void load_run(float (&v)[M], const float *p)
{
    if constexpr (M == 1) {
        v[0] = p[0];
    } else if constexpr (M == 2) {
        const f2u t = *reinterpret_cast<const f2u *>(p);
        v[0] = t.x; v[1] = t.y;
    } else {
#pragma unroll
        for (int n = 0; n < M; n += 4) {
            const f4u t = *reinterpret_cast<const f4u *>(p + n);
            v[n] = t.x; v[n + 1] = t.y; v[n + 2] = t.z; v[n + 3] = t.w;
        }
    }
}

MFL_D bool wave_any(bool pred)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_ballot_w64(pred) != 0;   // wave-uniform
#else
    return pred;
#endif
}

// ---- thread -> (tile, t_0) mapping --------------------------------------------
// A wave's 64 lanes cover kWaveTiles consecutive tiles x kWaveR consecutive t_0.  Lanes of
// consecutive tiles gather from consecutive groups of M source pixels and lanes of
// consecutive t_0 from consecutive taps of the same padded row, so one wave-wide gather
// reads a few contiguous row pieces of about M*kWaveTiles + k floats instead of 64/k short
// ones.  The 4 waves of a 256-thread block take consecutive t_0 groups of the same tiles.
#ifndef MFL_WAVE_TILES
#define MFL_WAVE_TILES 4
#endif
constexpr int kWaveTiles = MFL_WAVE_TILES;
constexpr int kWaveR = 64 / kWaveTiles;
constexpr int kBlockWaves = 4;
constexpr int kBlockR = kWaveR * kBlockWaves;

MFL_HD int blocks_per_image(const Geom &g)
{
    return ((g.ntile + kWaveTiles - 1) / kWaveTiles) * ((g.CK + kBlockR - 1) / kBlockR);
}

// Per-thread column set (see header comment).
template <int M>
struct Columns {
    int l0;          // first column, clamped into [0, L)
    int lcl[M];      // column n clamped into [0, L)
    int ibase[M];    // y_n*Wp + x_n: top-left corner of column n's window in a padded plane
    bool valid[M];   // column n exists and this thread is not a grid-tail duplicate
};

// block = blockIdx.x within the image, tid = threadIdx.x (0..255)
template <int M>
MFL_D void make_columns(Columns<M> &col, const Geom &g, int block, int tid)
{
    const int nrg = (g.CK + kBlockR - 1) / kBlockR;
    const int tg = block / nrg, rg = block - tg * nrg;
    const int wave = tid >> 6, lane = tid & 63;
    const int i_raw = tg * kWaveTiles + lane / kWaveR;
    const int t_raw = rg * kBlockR + wave * kWaveR + (lane % kWaveR);
    const bool active = i_raw < g.ntile && t_raw < g.CK;
    const int i = i_raw < g.ntile ? i_raw : g.ntile - 1;
    const int t0 = t_raw < g.CK ? t_raw : g.CK - 1;
    const int l0 = i * (M * g.CK) + t0;
    col.l0 = l0 < g.L ? l0 : g.L - 1;
#pragma unroll
    for (int n = 0; n < M; ++n) {
        const int ln = l0 + n * g.CK;
        col.valid[n] = active && ln < g.L;
        const int lc = ln < g.L ? ln : g.L - 1;
        const int y = lc / g.W;
        col.lcl[n] = lc;
        col.ibase[n] = y * g.Wp + (lc - y * g.W);
    }
}

// acc[cc][n] += sum over the K taps of w[n][t] * S[c0+cc][l_n][t]   (cc < nch)
//   xp_b  zero-padded planes of image b: [C][Hp][Wp] (+ slack)
//   ip_b  zero-padded RGB image of image b as F4 per pixel: [Hp][Wp]   (FROM_CD = false)
//   cd_b  colour-distance tensor of image b: [L][K]                     (FROM_CD = true)
//   tab   CK-entry offset table (LDS or global)
// A Sink with enabled = true (the backward kernels) additionally receives
//   scatter(a, x, cc, w)   once per (tap, channel): the run address and the tap weights,
//   tap(t, pd, cdv, u)     once per tap, u[n] = sum_cc dmul[cc][n] * S[c0+cc][l_n][t].
struct NoSink {
    static constexpr bool enabled = false;
};

// Channels gathered per batch: all addresses of a batch are formed first, then all its loads are
// issued back to back, then the FMAs.  Straight-line phases keep a batch's loads in flight together.
MFL_HD constexpr int batch_width(int CC, int M) { return M >= 4 ? (CC < 7 ? CC : 7) : (CC < 8 ? CC : 8); }

template <int CC, int M, bool FROM_CD, bool EXACT, class TabPtr, class Sink>
MFL_D void window_sums(float (&acc)[CC][M], const Geom &g, const float *__restrict__ xp_b,
                       const F4 *__restrict__ ip_b, const float *__restrict__ cd_b, TabPtr tab, float pw, float cw,
                       int c0, int nch, const Columns<M> &col, const float (*dmul)[M], Sink &sink)
{
    constexpr int CB = batch_width(CC, M);
    const int tmp = c0 * g.rhoK + col.l0;
    const int lp0 = c0 * g.qK + tmp / g.CK;
    const int W = g.W, gap = g.Wp - g.W;
    // decode state of (channel c0, tap t): table index rs, source column xs, padded offset ls = y'*Wp + x'
    int rs = tmp % g.CK;
    int xs = lp0 % W;
    int ls = (lp0 / W) * g.Wp + xs;
    const int dq_lin = g.dqy * g.Wp + g.dqx, dK_lin = g.dKy * g.Wp + g.dKx;

    F4 ctr[M];
    if constexpr (!FROM_CD) {
        const int ctr_off = g.p0 * g.Wp + g.p0;
#pragma unroll
        for (int n = 0; n < M; ++n) ctr[n] = ip_b[col.ibase[n] + ctr_off];
    }

    int t = 0;
    for (int ky = 0; ky < g.k; ++ky) {
        for (int kx = 0; kx < g.k; ++kx, ++t) {
            const int dy = ky - g.p0, dx = kx - g.p0;
            const float pd = (float)(dy * dy + dx * dx);
            const float pdw = pd * pw;
            float w[M], cdv[M];
#pragma unroll
            for (int n = 0; n < M; ++n) {
                if constexpr (FROM_CD) {
                    cdv[n] = cd_b[(size_t)col.lcl[n] * g.K + t];
                } else {
                    const F4 nb = ip_b[col.ibase[n] + ky * g.Wp + kx];
                    const float d0 = ctr[n].x - nb.x, d1 = ctr[n].y - nb.y, d2 = ctr[n].z - nb.z;
                    cdv[n] = d0 * d0 + d1 * d1 + d2 * d2;
                }
                w[n] = fmaf(cdv[n], cw, pdw);
            }
            float u[M];
#pragma unroll
            for (int n = 0; n < M; ++n) u[n] = 0.0f;

            int r = rs, x = xs, lin = ls;
            int a_first = 0;
#pragma unroll
            for (int cb = 0; cb < CC; cb += CB) {
                int a[CB], xc[CB];
                bool cross = false;
                // phase A: addresses (and the decode step to the next channel)
#pragma unroll
                for (int i = 0; i < CB; ++i) {
                    if (cb + i < CC) {
                        int ai = lin + tab[r];
                        if (cb + i == 0) a_first = ai;
                        if constexpr (!EXACT) ai = (cb + i < nch) ? ai : a_first;   // channels past the chunk: a valid dummy
#ifdef MFL_DBG_SMALL_REGION
                        ai &= 0x3ff;   // timing experiment only: every gather hits the same 4 KB
#endif
                        a[i] = ai;
                        xc[i] = x;
                        if constexpr (M > 1) cross |= (x + M > W);
#ifdef MFL_DBG_TRIVIAL_STEP
                        lin += dK_lin;   // timing experiment only: no carries, wrong results
#else
                        r += g.rhoK;
                        const int carry = r >= g.CK;
                        r -= carry ? g.CK : 0;
                        x += g.dKx + carry;
                        const int wrap = x >= W;
                        x -= wrap ? W : 0;
                        lin += dK_lin + carry + (wrap ? gap : 0);
#endif
                    }
                }
                // phase B: the batch's loads.  A run that crosses the end of an image row continues
                // `gap` floats later; that is rare, so ONE wave-uniform branch per batch selects between
                // the all-wide form and a form in which the crossing lanes load their floats one by one.
                float v[CB][M];
#ifdef MFL_DBG_NO_LOADS
                if (true) {   // timing experiment only: no gather loads, wrong results
#pragma unroll
                    for (int i = 0; i < CB; ++i)
                        if (cb + i < CC)
#pragma unroll
                            for (int n = 0; n < M; ++n) v[i][n] = __int_as_float(a[i] + n);
                } else
#endif
                if (M == 1 || !wave_any(cross)) {
#pragma unroll
                    for (int i = 0; i < CB; ++i)
                        if (cb + i < CC) load_run<M>(v[i], xp_b + a[i]);
                } else {
#pragma unroll
                    for (int i = 0; i < CB; ++i)
                        if (cb + i < CC) {
                            if (xc[i] + M > W) {
#pragma unroll
                                for (int n = 0; n < M; ++n) v[i][n] = xp_b[a[i] + n + ((xc[i] + n >= W) ? gap : 0)];
                            } else {
                                load_run<M>(v[i], xp_b + a[i]);
                            }
                        }
                }
                // phase D: accumulate
#pragma unroll
                for (int i = 0; i < CB; ++i)
                    if (cb + i < CC) {
#pragma unroll
                        for (int n = 0; n < M; ++n) acc[cb + i][n] = fmaf(w[n], v[i][n], acc[cb + i][n]);
                        if constexpr (Sink::enabled) {
#pragma unroll
                            for (int n = 0; n < M; ++n) u[n] = fmaf(dmul[cb + i][n], v[i][n], u[n]);
                            if (EXACT || cb + i < nch) sink.scatter(a[i], xc[i], cb + i, w);
                        }
                    }
            }
            if constexpr (Sink::enabled) sink.tap(t, pd, cdv, u);

            // decode step to the next tap
            rs += g.rho;
            const int carry = rs >= g.CK;
            rs -= carry ? g.CK : 0;
            xs += g.dqx + carry;
            const int wrap = xs >= W;
            xs -= wrap ? W : 0;
            ls += dq_lin + carry + (wrap ? gap : 0);
        }
    }
}

#if defined(__HIPCC__)
// Order-free accumulation of the backward scatter: a contribution v is added as the integer
// round(v * scale) with a 64-bit integer atomic (integer sums do not depend on the order of the adds, so the
// gradient is reproducible bit for bit; float atomics are not).  scale is a power of two chosen by the
// host side so that |v * scale| < 2^50 for every contribution: 1.5 * 2^52 + v * scale is then exact in
// double and its low mantissa bits are the rounded integer in two's complement.
__device__ __forceinline__ void fixed_point_add(unsigned long long *p, float v, float scale)
{
    const double d = __builtin_fma((double)v, (double)scale, 6755399441055744.0);
    const long long q = __builtin_bit_cast(long long, d) - 0x4338000000000000LL;
    atomicAdd(p, (unsigned long long)q);
}
#endif

}  // namespace mfl

#endif  // MFL_CORE_H
