// Diagonal gather kernels of the local-window message-passing library (gfx950): the fast path
// for images at least 64 + k pixels wide.  Ownership map, address table and the reason a thread
// owns a diagonal of k output pixels: mfl_diag.h.  One translation unit per (CH, KS):
// compile with -DMFL_DIAG_CH=<8|10|11> -DMFL_DIAG_KS=<3|5|7|9|11>.
//
// One workgroup = NGW consecutive groups x NCS channel splits (one wave each) of one b-block of one
// image.  Per weight group (up to TB taps of one tap column):
//   weight phase  every thread computes the tap weights pd*pos_w + col_w*|I(l) - I(l + tap)|^2 of
//                 its share of the workgroup's pixels (lanes = consecutive pixels, so the image
//                 reads are coalesced) into an LDS table, double buffered: one barrier per group;
//   gather phase  every wave reads the weights of its own pixels back from LDS and, per channel,
//                 loads the shared elements A and B (one coalesced dword per lane each, scalar
//                 offsets straight from the table: no per-lane address arithmetic) for k FMAs.
// Epilogue: the accumulators go through LDS so that threads own whole pixels again (all channels,
// consecutive lanes = consecutive pixels) for the fused normalise step or the plain product.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfl.h"
#include "mfl_diag_api.h"

#if !defined(MFL_DIAG_CH) || !defined(MFL_DIAG_KS)
#error "compile with -DMFL_DIAG_CH=<channels per wave> -DMFL_DIAG_KS=<window side>"
#endif

using namespace mfl;

namespace {

template <int CH, int KS, int NGW, int NCS>
struct DiagCfg {
    static constexpr int NT = 64 * NGW * NCS;
    static constexpr int RUN = NGW * KS;
    static constexpr int RUNP = RUN | 1;
    static constexpr int NBR = 64 + KS - 1;
    static constexpr int NSLOT = NBR * RUN;
    static constexpr int S = (NSLOT + NT - 1) / NT;
    static constexpr int SLOTP = NBR * RUNP;
    static constexpr int CE = CH * NCS;
};

__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, int voff, uint32_t soff)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, (int)soff, 0));
}

// Per-thread state that lives through the whole kernel.
template <int CH, int KS, int NGW, int NCS>
struct DiagState {
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    float acc[CH][KS];
    int ibase[Cfg::S];          // top-left corner of the pixel's window in the padded image
    float cr[Cfg::S], cg[Cfg::S], cb[Cfg::S];
};

// ---- weight phase: tap weights of one group into wt[tt][slot] ------------------------------
template <int CH, int KS, int NGW, int NCS>
__device__ __forceinline__ void weight_phase(const DiagState<CH, KS, NGW, NCS> &st, const DiagGeom &g,
                                             const F4 *__restrict__ ip_b, float pw, float cw, uint32_t gw, float *wt)
{
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    const int kx = (gw >> 8) & 255, ky0 = (gw >> 16) & 255, nky = gw >> 24;
    const int tid = threadIdx.x;
    const int dx = kx - g.p0;
    for (int tt = 0; tt < nky; ++tt) {
        const int ky = ky0 + tt;
        const int dy = ky - g.p0;
        const float pdw = (float)(dy * dy + dx * dx) * pw;
        const int toff = ky * g.Wp + kx;
        F4 nb[Cfg::S];
#pragma unroll
        for (int i = 0; i < Cfg::S; ++i) nb[i] = ip_b[st.ibase[i] + toff];
#pragma unroll
        for (int i = 0; i < Cfg::S; ++i) {
            const int sigma = tid + i * Cfg::NT;
            const int b_rel = sigma / Cfg::RUN, a_rel = sigma - b_rel * Cfg::RUN;
            const float d0 = st.cr[i] - nb[i].x, d1 = st.cg[i] - nb[i].y, d2 = st.cb[i] - nb[i].z;
            const float cd = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
            if (Cfg::NSLOT % Cfg::NT == 0 || sigma < Cfg::NSLOT) wt[tt * Cfg::SLOTP + b_rel * Cfg::RUNP + a_rel] = fmaf(cd, cw, pdw);
        }
    }
}

// Table record of one tap for one wave, held in scalar registers.
template <int CH>
struct TapRec {
    uint32_t eA[CH], eB[CH], flags;
};

template <int CH, bool WITH_B>
__device__ __forceinline__ void load_rec(TapRec<CH> &r, const uint32_t *__restrict__ rec)
{
#pragma unroll
    for (int i = 0; i < CH; ++i) r.eA[i] = rec[i];
    if constexpr (WITH_B) {
#pragma unroll
        for (int i = 0; i < CH; ++i) r.eB[i] = rec[CH + i];
    }
    r.flags = rec[2 * CH];
}

// ---- gather phase of one group, specialised on the alignment SA of its tap column ------------
// Every step first runs the unflagged form (lane offsets lane*4, all pixels share A or B).  Steps
// that meet an image-row start are then corrected in place: lanes behind the row start read gap
// floats further (the padding between rows), and pixels whose source pixel is still on the previous
// row read gap floats earlier than their lane's shared element.
template <int CH, int KS, int NGW, int NCS, int SA>
__device__ __forceinline__ void gather_phase(DiagState<CH, KS, NGW, NCS> &st, const DiagGeom &g,
                                             __amdgpu_buffer_rsrc_t rsrc, const uint32_t *__restrict__ table,
                                             size_t rec0, uint32_t gw, const float *wt, int gi, int lane)
{
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    const int kx = (gw >> 8) & 255, ky0 = (gw >> 16) & 255, nky = gw >> 24;
    const int voff = lane * 4;
    const int wbase = (lane + KS - 1) * Cfg::RUNP + gi * KS;
    const size_t trec = (size_t)g.NCS * g.RECW;
    const uint32_t *__restrict__ rec = table + rec0 + (size_t)(ky0 * KS + kx) * trec;
    const size_t rstep = (size_t)KS * trec;
    TapRec<CH> cur;
    load_rec<CH, (SA > 0)>(cur, rec);
    for (int tt = 0; tt < nky; ++tt) {
        float A[CH], Bv[CH];
#pragma unroll
        for (int i = 0; i < CH; ++i) A[i] = buf_load(rsrc, voff, cur.eA[i]);
        if constexpr (SA > 0) {
#pragma unroll
            for (int i = 0; i < CH; ++i) Bv[i] = buf_load(rsrc, voff, cur.eB[i]);
        }
        float w[KS];
#pragma unroll
        for (int n = 0; n < KS; ++n) w[n] = wt[tt * Cfg::SLOTP + wbase - n * (Cfg::RUNP - 1)];
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int n = 0; n < KS; ++n) st.acc[i][n] = fmaf(w[n], (n < KS - SA) ? A[i] : Bv[i], st.acc[i][n]);

        if (cur.flags != 0) {
            const int W = g.W, gap4 = g.gap * 4;
            uint32_t xw[(CH + 1) / 2];
#pragma unroll
            for (int i = 0; i < (CH + 1) / 2; ++i) xw[i] = rec[2 * CH + 1 + i];
#pragma unroll
            for (int i = 0; i < CH; ++i) {
                if ((cur.flags >> i) & 1) {
                    const uint32_t x16 = (xw[i / 2] >> (16 * (i & 1))) & 0xffffu;
                    const int xA = x16 & 0x7fff;
                    int pA = xA + lane;
                    const bool wrA = pA >= W;
                    pA -= wrA ? W : 0;
                    const int vA = voff + (wrA ? gap4 : 0);
                    const float A1 = buf_load(rsrc, vA, cur.eA[i]) - A[i];
                    const float X1 = buf_load(rsrc, vA, cur.eA[i] - (uint32_t)gap4) - A[i];
                    float B1 = 0.f, Y1 = 0.f;
                    int pB = 0;
                    if constexpr (SA > 0) {
                        const int xB = (x16 & 0x8000) ? (xA + 1 == W ? 0 : xA + 1) : xA;
                        pB = xB + lane;
                        const bool wrB = pB >= W;
                        pB -= wrB ? W : 0;
                        const int vB = voff + (wrB ? gap4 : 0);
                        B1 = buf_load(rsrc, vB, cur.eB[i]) - Bv[i];
                        Y1 = buf_load(rsrc, vB, cur.eB[i] - (uint32_t)gap4) - Bv[i];
                    }
#pragma unroll
                    for (int n = 0; n < KS; ++n) {
                        float dv;
                        if (n < KS - SA) dv = (pA < n) ? X1 : A1;
                        else dv = (pB < n) ? Y1 : B1;
                        st.acc[i][n] = fmaf(w[n], dv, st.acc[i][n]);
                    }
                }
            }
        }
        rec += rstep;
        if (tt + 1 < nky) load_rec<CH, (SA > 0)>(cur, rec);
    }
}

template <int CH, int KS, int NGW, int NCS>
struct Loop {
    DiagState<CH, KS, NGW, NCS> &st;
    const DiagGeom &g;
    __amdgpu_buffer_rsrc_t rsrc;
    const uint32_t *__restrict__ table;
    const uint32_t *__restrict__ groups;
    const F4 *__restrict__ ip_b;
    float *lds;
    size_t rec0;
    float pw, cw;
    int gi, lane, gidx;
    bool live;

    // all weight groups of alignment class SA, then the next class
    template <int SA>
    __device__ __forceinline__ void classes()
    {
        using Cfg = DiagCfg<CH, KS, NGW, NCS>;
        const int ngroups = g.ngroups;
        const int bufsz = g.TB * Cfg::SLOTP;
        while (gidx < ngroups) {
            const uint32_t gw = groups[gidx];
            if ((int)(gw & 255) != SA) break;
            if (gidx + 1 < ngroups)
                weight_phase<CH, KS, NGW, NCS>(st, g, ip_b, pw, cw, groups[gidx + 1], lds + ((gidx + 1) & 1) * bufsz);
            if (live)
                gather_phase<CH, KS, NGW, NCS, SA>(st, g, rsrc, table, rec0, gw, lds + (gidx & 1) * bufsz, gi, lane);
            __syncthreads();
            ++gidx;
        }
        if constexpr (SA + 1 < KS) classes<SA + 1>();
    }
};

// MODE 0: fused mean-field update   MODE 1: y = x * window sums
template <int CH, int KS, int NGW, int NCS, int MODE>
__global__ void __launch_bounds__(64 * NGW * NCS)
diag_kernel(DiagGeom g, int epi_rows, const float *__restrict__ xp, const F4 *__restrict__ ip4,
            const uint32_t *__restrict__ table, const uint32_t *__restrict__ groups, const float *__restrict__ pos_w,
            const float *__restrict__ col_w, const float *__restrict__ unary, const float *__restrict__ aw,
            const float *__restrict__ ab, const float *__restrict__ compat, float *__restrict__ logq,
            float *__restrict__ qp_out, float *__restrict__ y_out)
{
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave % NGW, ci = wave / NGW;
    const int img = blockIdx.x % g.B;
    const int tile = blockIdx.x / g.B;
    const int gb = tile % g.NGB, beta = tile / g.NGB;
    const int gamma = gb * NGW + gi;
    const bool live = gamma < g.NG;
    const int C = g.C, CK = g.CK, L = g.L, W = g.W, Wp = g.Wp;

    const size_t img_off = (size_t)img * C * g.plane;
    const float *in_b = xp + img_off;                 // FS floats before the image's padded planes
    const F4 *__restrict__ ip_b = ip4 + (size_t)img * g.plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in_b, 0, 0x7fffffff, 0x00020000);
    const float pw = *pos_w, cw = *col_w;
    const int ctr_off = g.p0 * Wp + g.p0;

    DiagState<CH, KS, NGW, NCS> st;
#pragma unroll
    for (int i = 0; i < CH; ++i)
#pragma unroll
        for (int n = 0; n < KS; ++n) st.acc[i][n] = 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::S; ++i) {
        const int sigma = tid + i * Cfg::NT;
        const int sc = sigma < Cfg::NSLOT ? sigma : 0;
        const int b_rel = sc / Cfg::RUN, a_rel = sc - b_rel * Cfg::RUN;
        const int aa = gb * Cfg::RUN + a_rel, bb = 64 * beta + b_rel - (KS - 1);
        const long long l = (long long)aa + (long long)bb * CK;
        const bool valid = bb >= 0 && aa < CK && l < L;
        const int lc = valid ? (int)l : 0;
        const int y = lc / W, x = lc - y * W;
        st.ibase[i] = y * Wp + x;
        const F4 c = ip_b[st.ibase[i] + ctr_off];
        st.cr[i] = c.x; st.cg[i] = c.y; st.cb[i] = c.z;
    }

    Loop<CH, KS, NGW, NCS> loop{st, g, rsrc, table, groups, ip_b, lds,
                                diag_record_index(g, beta, live ? gamma : 0, 0, ci), pw, cw, gi, lane, 0, live};
    weight_phase<CH, KS, NGW, NCS>(st, g, ip_b, pw, cw, groups[0], lds);
    __syncthreads();
    loop.template classes<0>();

    // ---- epilogue: accumulators -> LDS [channel][b_rel][a_rel], then one pixel per thread ----
    const int CBR = epi_rows;
    for (int cb0 = 0; cb0 < Cfg::NBR; cb0 += CBR) {
        if (live) {
#pragma unroll
            for (int n = 0; n < KS; ++n) {
                const int br = lane - n + KS - 1 - cb0;
                if (br >= 0 && br < CBR) {
#pragma unroll
                    for (int i = 0; i < CH; ++i)
                        if (ci * CH + i < C) lds[((ci * CH + i) * CBR + br) * Cfg::RUNP + gi * KS + n] = st.acc[i][n];
                }
            }
        }
        __syncthreads();
        const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
        const int cnt = rows * Cfg::RUN;
        for (int r = tid; r < cnt; r += Cfg::NT) {
            const int br = r / Cfg::RUN, a_rel = r - br * Cfg::RUN;
            const int b_rel = cb0 + br;
            const int n = a_rel % KS;
            const int owner = b_rel - (KS - 1) + n;
            const int aa = gb * Cfg::RUN + a_rel, bb = 64 * beta + b_rel - (KS - 1);
            const long long l64 = (long long)aa + (long long)bb * CK;
            if (owner < 0 || owner > 63 || bb < 0 || aa >= CK || l64 >= L) continue;
            const int l = (int)l64;
            const int y = l / W, x = l - y * W;
            const int cpos = (y + g.p0) * Wp + x + g.p0;
            const float *msg = lds + br * Cfg::RUNP + a_rel;
            const int cstride = CBR * Cfg::RUNP;
            const float *qc = in_b + g.FS + cpos;
            if constexpr (MODE == 1) {
                float *y_b = y_out + (size_t)img * C * L + l;
                for (int c = 0; c < C; ++c) y_b[(size_t)c * L] = msg[c * cstride] * qc[(size_t)c * g.plane];
            } else {
                constexpr int CE = Cfg::CE;
                const float *u_b = unary + (size_t)img * C * L + l;
                float m[CE], u[CE];
                float mu = -INFINITY;
#pragma unroll
                for (int c = 0; c < CE; ++c)
                    if (c < C) {
                        float v = msg[c * cstride] * qc[(size_t)c * g.plane];
                        if (aw) v = fmaf(aw[(size_t)c * L + l], v, ab[(size_t)c * L + l]);
                        m[c] = v;
                        u[c] = u_b[(size_t)c * L];
                        mu = fmaxf(mu, u[c]);
                    }
                float su = 0.f;
#pragma unroll
                for (int c = 0; c < CE; ++c)
                    if (c < C) su += __expf(u[c] - mu);
                const float lse_u = mu + __logf(su);
                float mv = -INFINITY;
#pragma unroll
                for (int c = 0; c < CE; ++c)
                    if (c < C) {
                        float p;
                        if (compat) {
                            p = 0.f;
#pragma unroll
                            for (int c2 = 0; c2 < CE; ++c2)
                                if (c2 < C) p = fmaf(compat[c * C + c2], m[c2], p);
                        } else {
                            p = m[c];
                        }
                        u[c] = (u[c] - lse_u) - p;
                        mv = fmaxf(mv, u[c]);
                    }
                float sv = 0.f;
#pragma unroll
                for (int c = 0; c < CE; ++c)
                    if (c < C) sv += __expf(u[c] - mv);
                const float lse_v = mv + __logf(sv);
                float *qo = qp_out + img_off + g.FS + cpos;
#pragma unroll
                for (int c = 0; c < CE; ++c)
                    if (c < C) {
                        const float lq = u[c] - lse_v;
                        if (logq) logq[((size_t)img * C + c) * L + l] = lq;
                        qo[(size_t)c * g.plane] = __expf(lq);
                    }
            }
        }
        __syncthreads();
    }
}

__global__ void diag_table_kernel(DiagGeom g, uint32_t *__restrict__ table)
{
    const size_t nrec = (size_t)g.NBB * g.NG * g.K * g.NCS;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(r % g.NCS);
        size_t rest = r / g.NCS;
        const int t = (int)(rest % g.K);
        rest /= g.K;
        const int gamma = (int)(rest % g.NG);
        const int beta = (int)(rest / g.NG);
        diag_fill_record(g, beta, gamma, t, ci, table + r * g.RECW);
    }
}

__global__ void diag_groups_kernel(DiagGeom g, uint32_t *__restrict__ groups)
{
    if (blockIdx.x == 0 && threadIdx.x == 0) diag_groups(g, groups);
}

template <int CH, int KS, int NGW, int NCS>
int launch_cfg(const DiagArgs &a, int mode, hipStream_t st)
{
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    const DiagGeom &g = a.g;
    if (g.CH != CH || g.k != KS || g.NGW != NGW || g.NCS != NCS) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)((size_t)g.NBB * g.NGB * g.B));
    const size_t lds = a.lds_bytes;
    hipError_t e;
    if (mode == 0) {
        auto kern = diag_kernel<CH, KS, NGW, NCS, 0>;
        e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), lds, st, a.g, a.epi_rows, a.xp, a.ip4, a.table, a.groups, a.pos_w, a.col_w,
                           a.unary, a.aw, a.ab, a.compat, a.logq, a.qp_out, a.y);
    } else {
        auto kern = diag_kernel<CH, KS, NGW, NCS, 1>;
        e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), lds, st, a.g, a.epi_rows, a.xp, a.ip4, a.table, a.groups, a.pos_w, a.col_w,
                           a.unary, a.aw, a.ab, a.compat, a.logq, a.qp_out, a.y);
    }
    return (int)hipGetLastError();
}

}  // namespace

namespace mfl {

#define MFL_CAT3_(a, b, c, d) a##b##c##d
#define MFL_CAT3(a, b, c, d) MFL_CAT3_(a, b, c, d)

int MFL_CAT3(launch_diag_ch, MFL_DIAG_CH, _k, MFL_DIAG_KS)(const DiagArgs &a, int mode, hipStream_t st)
{
    constexpr int CH = MFL_DIAG_CH, KS = MFL_DIAG_KS;
    if constexpr (CH == 8) {
        if (a.g.NCS == 1) return launch_cfg<CH, KS, 8, 1>(a, mode, st);
        if (a.g.NCS == 2) return launch_cfg<CH, KS, 4, 2>(a, mode, st);
        return launch_cfg<CH, KS, 2, 3>(a, mode, st);
    } else {
        if (a.g.NCS == 2) return launch_cfg<CH, KS, 4, 2>(a, mode, st);
        if (a.g.NCS == 3) return launch_cfg<CH, KS, 2, 3>(a, mode, st);
        return MFL_ERR_INVALID_ARGUMENT;
    }
}

#if MFL_DIAG_CH == 8 && MFL_DIAG_KS == 3
int launch_diag_table(const DiagGeom &g, uint32_t *table, uint32_t *groups, hipStream_t st)
{
    hipLaunchKernelGGL(diag_groups_kernel, dim3(1), dim3(64), 0, st, g, groups);
    const size_t nrec = (size_t)g.NBB * g.NG * g.K * g.NCS;
    const unsigned blocks = (unsigned)((nrec + 255) / 256 < 65535u * 8 ? (nrec + 255) / 256 : 65535u * 8);
    hipLaunchKernelGGL(diag_table_kernel, dim3(blocks), dim3(256), 0, st, g, table);
    return (int)hipGetLastError();
}
#endif

}  // namespace mfl
