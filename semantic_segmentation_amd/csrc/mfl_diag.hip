// Diagonal gather kernels of the local-window message-passing library (gfx950): the fast path
// for images at least 64 + k pixels wide.  Ownership map, address table and the reason a thread
// owns a diagonal of k output pixels: mfl_diag.h.  One translation unit per (CH, KS):
// compile with -DMFL_DIAG_CH=<8|10|11> -DMFL_DIAG_KS=<3|5|7>, or -DMFL_DIAG_CH=5 -DMFL_DIAG_KS=<9|11>.
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
// The backward reuses the machinery transposed: D = dy * x enters through LDS into the accumulator
// registers, and per (channel, tap) a lane sums what its k pixels send to the shared element and
// stores the sum into the copy of the gradient planes that belongs to (side, window row of the element's
// line): every element has one writer per copy, so no atomic is needed and the 2k copies are added in a
// fixed order afterwards (tap_scatter).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>

#include "../../include/mfl.h"
#include "mfl_diag_api.h"
#include "mfl_tail.h"

#if !defined(MFL_DIAG_CH) || !defined(MFL_DIAG_KS)
#error "compile with -DMFL_DIAG_CH=<channels per wave> -DMFL_DIAG_KS=<window side>"
#endif

using namespace mfl;

namespace {

template <int CH, int KS, int NGW, int NCS>
struct DiagCfg {
    static constexpr int NT = 64 * NGW * NCS;
    static constexpr int RUN = NGW * KS;
    static constexpr int RUNP = (RUN & 1) ? RUN + 2 : RUN + 1;
    static constexpr int NBR = 64 + KS - 1;
    static constexpr int NSLOT = NBR * RUN;
    static constexpr int S = (NSLOT + NT - 1) / NT;
    static constexpr int SLOTP = NBR * RUNP;
    static_assert(RUNP > RUN, "the last padded column of the weight table serves as a spare place");
    static constexpr int CE = CH * NCS;
};

#ifdef MFL_STAMPS
// diagnostic build only: thread 0 of the first 256 workgroups records cycle stamps
#define MFL_STAMP(i) do { if (dbg && blockIdx.x < 256 && threadIdx.x == 0) dbg[blockIdx.x * 64 + (i)] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFL_STAMP(i) do { } while (0)
#endif

// MFL_ABL_* macros: timing experiments only (tools/abl_diag.sh); they remove one part of the kernel
// and give wrong results.
__device__ __forceinline__ float buf_load(__amdgpu_buffer_rsrc_t r, int voff, uint32_t soff)
{
#ifdef MFL_ABL_NO_LOADS
    return __builtin_bit_cast(float, (uint32_t)voff + soff);
#endif
#ifdef MFL_ABL_SMALL_WINDOW
    soff &= MFL_ABL_SMALL_WINDOW;    // timing experiment: every element load inside one small window
#endif
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, (int)soff, 0));
}

// Per-thread state that lives through the whole kernel.
template <int CH, int KS, int NGW, int NCS>
struct DiagState {
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    float acc[CH][KS];
    int ioff[Cfg::S];           // byte offset of the top-left corner of the pixel's window in the padded image
    int lidx[Cfg::S];           // the pixel's place in a tap's LDS weight table (a spare place past the table
                                // for the thread slots beyond the workgroup's pixels)
    float cr[Cfg::S], cg[Cfg::S], cb[Cfg::S];
};

// ---- weight phase: tap weights of one group into wt[tt][slot] ------------------------------
// Software pipelined over the taps of the group: the image loads of tap tt+1 are in flight while
// the weights of tap tt are computed and stored.
template <int CH, int KS, int NGW, int NCS>
struct WeightPhase {
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    struct Nb { float r[Cfg::S], g[Cfg::S], b[Cfg::S]; };

    // irsrc: buffer descriptor of the image's padded {r,g,b,0} pixels; toff: tap offset in pixels
    static __device__ __forceinline__ void load(Nb &nb, const DiagState<CH, KS, NGW, NCS> &st,
                                                __amdgpu_buffer_rsrc_t irsrc, int toff)
    {
#pragma unroll
        for (int i = 0; i < Cfg::S; ++i) {
            typedef float f3 __attribute__((ext_vector_type(3)));
            const f3 v = __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(irsrc, st.ioff[i], toff * 16, 0));
            nb.r[i] = v.x; nb.g[i] = v.y; nb.b[i] = v.z;
        }
    }
    static __device__ __forceinline__ void store(const Nb &nb, const DiagState<CH, KS, NGW, NCS> &st, float pdw,
                                                 float cw, float *wt_tap)
    {
#pragma unroll
        for (int i = 0; i < Cfg::S; ++i) {
            const float d0 = st.cr[i] - nb.r[i], d1 = st.cg[i] - nb.g[i], d2 = st.cb[i] - nb.b[i];
            const float cd = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
            wt_tap[st.lidx[i]] = fmaf(cd, cw, pdw);
        }
    }
    static __device__ __forceinline__ void run(const DiagState<CH, KS, NGW, NCS> &st, int p0, int Wp,
                                               __amdgpu_buffer_rsrc_t irsrc, float pw, float cw, uint32_t gw, float *wt)
    {
#ifdef MFL_ABL_NO_WEIGHT
        return;
#endif
        const int kx = (gw >> 8) & 255, ky0 = (gw >> 16) & 255, nky = gw >> 24;
        const int dx = kx - p0;
        auto pdw = [&](int tt) { const int dy = ky0 + tt - p0; return (float)(dy * dy + dx * dx) * pw; };
        const int toff0 = ky0 * Wp + kx;
        Nb n0, n1;
        load(n0, st, irsrc, toff0);
        int tt = 0;
        for (; tt + 1 < nky; tt += 2) {
            load(n1, st, irsrc, toff0 + (tt + 1) * Wp);
            store(n0, st, pdw(tt), cw, wt + tt * Cfg::SLOTP);
            if (tt + 2 < nky) load(n0, st, irsrc, toff0 + (tt + 2) * Wp);
            store(n1, st, pdw(tt + 1), cw, wt + (tt + 1) * Cfg::SLOTP);
        }
        if (tt < nky) store(n0, st, pdw(tt), cw, wt + tt * Cfg::SLOTP);
    }
};

// ---- gather phase ---------------------------------------------------------------------------
// The taps of an iteration form one stream (group-list order).  A wave keeps the table record of a
// tap in ONE vector register (lane j = dword j; layout: diag_recw in mfl_diag.h) and pulls the scalar
// offsets out of it with v_readlane, so records can be requested several taps ahead without holding
// scalar registers.  The loaded elements live in one register set that is refilled in place: as soon
// as the FMAs of step i have consumed A[i] (B[i]) the load of the next tap's A[i] (B[i]) is issued
// into the same register, which keeps about one tap of loads in flight at all times.
template <int CH>
struct TapVals {
    float A[CH], B[CH];
    // per slot (a step that meets an image-row start): the extra element of side A, of side B
    float EA[kDiagSlots], EB[kDiagSlots];
};

__device__ __forceinline__ uint32_t rec_word(uint32_t recv, int j)
{
    return (uint32_t)__builtin_amdgcn_readlane((int)recv, j);
}

// ---- corrections of flagged steps -------------------------------------------------------------
// A flagged side has a threshold T (record layout: mfl_diag.h): with p = lane - T, pixel n of the lane
// must read the "extra" element (gap floats after the table's "shared" one) if p >= n.  The taps
// accumulate w[n] * shared for every pixel; the correction adds w[n] * (extra - shared) where p >= n.
// The step a slot corrects is known at run time only, but the accumulators must be reached through
// constant register names.  A C++ branch per step around accumulator updates makes the compiler copy
// the accumulators on every path, so the whole decision sits inside ONE asm statement per slot: an
// the per-pixel selection of the weight (t[n] = p[n] >= n ? w[n] : 0), then a computed jump
// (s_getpc + step * body size) into a table of per-step bodies, each k times (v_sub: extra - shared
// element of that step, in-place v_fmac on its accumulator) and a branch out.  The bodies must
// keep the fixed size MFL_CORR_STRIDE = 8*k + 4 bytes (2k 4-byte VOP2 + s_branch).
#if MFL_DIAG_CH == 5
#define MFL_CORR_LIST(M) M(0) M(1) M(2) M(3) M(4)
#elif MFL_DIAG_CH == 8
#define MFL_CORR_LIST(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#elif MFL_DIAG_CH == 10
#define MFL_CORR_LIST(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9)
#elif MFL_DIAG_CH == 11
#define MFL_CORR_LIST(M) M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7) M(8) M(9) M(10)
#else
#error "no correction step list for this MFL_DIAG_CH"
#endif
#if MFL_DIAG_KS == 3
#define MFL_KS_LIST(M, I) M(I, 0) M(I, 1) M(I, 2)
#define MFL_CORR_STRIDE "28"
#define MFL_CORR_SELS MFL_CORR_SEL2(0, 1) MFL_CORR_SEL1(2)
#elif MFL_DIAG_KS == 5
#define MFL_KS_LIST(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3) M(I, 4)
#define MFL_CORR_STRIDE "44"
#define MFL_CORR_SELS MFL_CORR_SEL2(0, 1) MFL_CORR_SEL2(2, 3) MFL_CORR_SEL1(4)
#elif MFL_DIAG_KS == 7
#define MFL_KS_LIST(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3) M(I, 4) M(I, 5) M(I, 6)
#define MFL_CORR_STRIDE "60"
#define MFL_CORR_SELS MFL_CORR_SEL2(0, 1) MFL_CORR_SEL2(2, 3) MFL_CORR_SEL2(4, 5) MFL_CORR_SEL1(6)
#elif MFL_DIAG_KS == 9
#define MFL_KS_LIST(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3) M(I, 4) M(I, 5) M(I, 6) M(I, 7) M(I, 8)
#define MFL_CORR_STRIDE "76"
#define MFL_CORR_SELS MFL_CORR_SEL2(0, 1) MFL_CORR_SEL2(2, 3) MFL_CORR_SEL2(4, 5) MFL_CORR_SEL2(6, 7) MFL_CORR_SEL1(8)
#elif MFL_DIAG_KS == 11
#define MFL_KS_LIST(M, I) M(I, 0) M(I, 1) M(I, 2) M(I, 3) M(I, 4) M(I, 5) M(I, 6) M(I, 7) M(I, 8) M(I, 9) M(I, 10)
#define MFL_CORR_STRIDE "92"
#define MFL_CORR_SELS MFL_CORR_SEL2(0, 1) MFL_CORR_SEL2(2, 3) MFL_CORR_SEL2(4, 5) MFL_CORR_SEL2(6, 7) MFL_CORR_SEL2(8, 9) MFL_CORR_SEL1(10)
#else
#error "no pixel list for this MFL_DIAG_KS"
#endif
// common part: t[N] = (p[N] >= N) ? w[N] : 0 with p = lane - T of the pixel's side.  A VALU that
// reads a scalar register (VCC included) written by a VALU needs two wait states in between on
// gfx950, and nothing checks inline asm for that: two compares (into VCC and s[98:99]) and one
// s_nop precede the two selects, the last pixel waits with s_nop 1.
#define MFL_CORR_SEL2(N0, N1)                                                                      \
    "v_cmp_le_i32 vcc, " #N0 ", %[p" #N0 "]\n\tv_cmp_le_i32 s[98:99], " #N1 ", %[p" #N1 "]\n\ts_nop 0\n\t" \
    "v_cndmask_b32 %[t" #N0 "], 0, %[w" #N0 "], vcc\n\tv_cndmask_b32 %[t" #N1 "], 0, %[w" #N1 "], s[98:99]\n\t"
#define MFL_CORR_SEL1(N) \
    "v_cmp_le_i32 vcc, " #N ", %[p" #N "]\n\ts_nop 1\n\tv_cndmask_b32 %[t" #N "], 0, %[w" #N "], vcc\n\t"
// body of step I, pixel N: acc += t * (extra element of the pixel's side - its shared element of step I)
#define MFL_CORR_FMA(I, N) \
    "v_sub_f32_e32 %[d], %[e" #N "], %[x" #I "_" #N "]\n\tv_fmac_f32_e32 %[a" #I "_" #N "], %[t" #N "], %[d]\n\t"
#define MFL_CORR_BODY(I) MFL_KS_LIST(MFL_CORR_FMA, I) "s_branch 9f\n\t"
#define MFL_CORR_ACC(I, N) [a##I##_##N] "+v"(st.acc[I][N]),
#define MFL_CORR_ACCS(I) MFL_KS_LIST(MFL_CORR_ACC, I)
#define MFL_CORR_TMP(I, N) [t##N] "=&v"(t[N]),
#define MFL_CORR_IN(I, N) [w##N] "v"(w[N]), [p##N] "v"(N < NA ? pA : pB), [e##N] "v"(N < NA ? eA : eB),
#define MFL_CORR_X(I, N) [x##I##_##N] "v"(N < NA ? v.A[I] : v.B[I]),
#define MFL_CORR_XS(I) MFL_KS_LIST(MFL_CORR_X, I)
// s[96:97] hold the jump target: named registers because the two halves are added separately.
// The jump lands at (address after s_getpc) + 12 + step * MFL_CORR_STRIDE: the assembler checks that
// the three instructions before the first body take 12 bytes and that the bodies together take
// CH * MFL_CORR_STRIDE bytes (all bodies are the same instruction sequence), so that another encoding
// of one of these instructions fails the build instead of moving the landing point.
#define MFL_STR_(x) #x
#define MFL_STR(x) MFL_STR_(x)
#define MFL_CORR_SIZE_CHECKS                                                                              \
    ".if (8b - 7b) != 12\n\t.error \"slot correction: jump preamble is not 12 bytes\"\n\t.endif\n\t"           \
    ".if (9b - 8b) != " MFL_STR(MFL_DIAG_CH) " * " MFL_CORR_STRIDE "\n\t"                                    \
    ".error \"slot correction: step bodies do not have the size the computed jump assumes\"\n\t.endif\n\t"
#define MFL_CORR_SLOT()                                                                                   \
    asm volatile(MFL_CORR_SELS                                                                            \
                 "s_mul_i32 %[tmp], %[st], " MFL_CORR_STRIDE "\n\ts_add_u32 %[tmp], %[tmp], 12\n\t"      \
                 "s_getpc_b64 s[96:97]\n\t7:\n\ts_add_u32 s96, s96, %[tmp]\n\ts_addc_u32 s97, s97, 0\n\t" \
                 "s_setpc_b64 s[96:97]\n\t8:\n\t"                                                        \
                 MFL_CORR_LIST(MFL_CORR_BODY) "9:\n\t" MFL_CORR_SIZE_CHECKS                              \
                 : MFL_CORR_LIST(MFL_CORR_ACCS) MFL_KS_LIST(MFL_CORR_TMP, 0)[d] "=&v"(d), [tmp] "=&s"(tmp) \
                 : MFL_KS_LIST(MFL_CORR_IN, 0) MFL_CORR_LIST(MFL_CORR_XS)[st] "s"(step)                   \
                 : "vcc", "scc", "s96", "s97", "s98", "s99")

// BWD: the scatter of the backward pass (tap_scatter) instead of the gather (tap)
template <int CH, int KS, int NGW, int NCS, bool BWD>
struct Loop {
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    float *scatter_out;                     // BWD: FS floats before the image's padded planes in copy 0 of the gradient planes (zeroed)
    size_t copy_stride;                     // BWD: floats between consecutive copies
    bool edge;                              // BWD (wave-uniform): a lane of the wave is an edge lane
    int C, ci;                              // BWD: channels of the launch (a block of a larger label set, or all), this wave's channel split
    DiagState<CH, KS, NGW, NCS> &st;
    __amdgpu_buffer_rsrc_t rsrc;
    const uint32_t *__restrict__ table;    // records of this wave's (b-block, group, channel split), tap 0
    const uint32_t *__restrict__ groups;
    __amdgpu_buffer_rsrc_t irsrc;           // the image's padded {r,g,b,0} pixels
    float *lds;
    float pw, cw;
    int ngroups, TB, trec, recw;            // trec: dwords between the records of consecutive taps
    int p0, Wp, W, gap4;
    int gi, lane;
    int gvoff;                              // the lane's byte offset in an element run: lane * 4, or past the descriptor's range
                                            // for a lane without pixels (the load returns 0 and requests no cache line)
    bool live;
    unsigned long long *dbg;

    // ---- the tap stream ----
    int pg, ptt;             // stream position of the next record to request: group, tap within it
    uint32_t gw_cur;         // descriptor of group pg
    uint32_t rec_use;        // record of the tap being accumulated
    uint32_t rec_pre;        // record of the next tap: its elements are being loaded
    uint32_t rec_req;        // record of the tap after that, requested
    TapVals<CH> v;

    // Request the record at the stream position and advance.  Past the end of the stream the last
    // record is requested again: its loads are issued and never used, which keeps the tap body free
    // of branches.
    __device__ __forceinline__ uint32_t request_record()
    {
        const uint32_t gw = gw_cur;
        const int kx = (gw >> 8) & 255, ky0 = (gw >> 16) & 255;
        const uint32_t *ptr = table + (size_t)((ky0 + ptt) * KS + kx) * trec;
        const int nky = gw >> 24;
        if (ptt + 1 < nky) {
            ++ptt;
        } else if (pg + 1 < ngroups) {
            ptt = 0;
            ++pg;
            gw_cur = groups[pg];
        }
        return ptr[lane < recw ? lane : 0];
    }

    // Extra elements of the record's slots (steps known at run time: dynamic lane of the record).
    __device__ __forceinline__ void load_slots(uint32_t recv)
    {
        const int voff = gvoff;
        constexpr int sb = 2 * CH + 1;
        const int nslots = (rec_word(recv, 2 * CH) >> 16) & 7;
#pragma unroll
        for (int j = 0; j < kDiagSlots; ++j) {
            if (j < nslots) {
                const int step = rec_word(recv, sb + j) & 15;
                v.EA[j] = buf_load(rsrc, voff, rec_word(recv, step) + (uint32_t)gap4);
                v.EB[j] = buf_load(rsrc, voff, rec_word(recv, CH + step) + (uint32_t)gap4);
            }
        }
    }

    // One slot (or one step past the slots): pixels n < KS - SA belong to side A, the others to side B.
    template <int SA>
    __device__ __forceinline__ void slot_apply(const float (&w)[KS], int pA, int pB, float eA, float eB, int step)
    {
        constexpr int NA = KS - SA;
        float t[KS], d;
        uint32_t tmp;
        MFL_CORR_SLOT();
        (void)t;
        (void)d;
        (void)tmp;
    }

    // Steps of the accumulated tap that meet an image-row start: corrected in place from the slots'
    // elements; a record with more flagged steps than slots (narrow images) walks the others with
    // loads on demand.
    template <int SA>
    __device__ __forceinline__ void correct(uint32_t hdr, const float (&w)[KS])
    {
        const int nslots = (hdr >> 16) & 7;
        constexpr int sb = 2 * CH + 1;
#pragma unroll
        for (int j = 0; j < kDiagSlots; ++j) {
            if (j < nslots) {
                const uint32_t desc = rec_word(rec_use, sb + j);
                const int pA = lane - ((int)((desc >> 8) & 255) - 64), pB = lane - ((int)((desc >> 16) & 255) - 64);
                slot_apply<SA>(w, pA, pB, v.EA[j], SA > 0 ? v.EB[j] : 0.f, desc & 15);
            }
        }
        if (hdr & 0x80000000u) overflow_loop<SA>(hdr, w);
    }

    // The steps past the slots, in a wave-uniform loop: each derives its thresholds from the step's
    // column word, loads its extra elements on demand and goes through the same correction block.
    template <int SA>
    __device__ __forceinline__ void overflow_loop(uint32_t hdr, const float (&w)[KS])
    {
        const int voff = gvoff;
        uint32_t m = hdr & 0xffffu;
#pragma unroll
        for (int j = 0; j < kDiagSlots; ++j) m &= m - 1;
        while (m) {
            const int i = __builtin_ctz(m);
            m &= m - 1;
            const uint32_t x16 = (rec_word(rec_use, diag_xw_base(CH) + (i >> 1)) >> (16 * (i & 1))) & 0xffffu;
            const int xA = x16 & 0x7fff;
            const int xB = (x16 & 0x8000) ? (xA + 1 == W ? 0 : xA + 1) : xA;
            auto threshold = [&](int x0) { return x0 + 63 > W - 1 ? W - x0 : (x0 < KS - 1 ? -x0 : 64); };
            const float eA = buf_load(rsrc, voff, rec_word(rec_use, i) + (uint32_t)gap4);
            float eB = 0.f;
            if constexpr (SA > 0) eB = buf_load(rsrc, voff, rec_word(rec_use, CH + i) + (uint32_t)gap4);
            slot_apply<SA>(w, lane - threshold(xA), lane - threshold(xB), eA, eB, i);
        }
    }

#ifdef MFL_STAMPS
    unsigned long long seg[6], seg_t;
#define MFL_SEG(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[i] += t_ - seg_t; seg_t = t_; } while (0)
#else
#define MFL_SEG(i) do { } while (0)
#endif
    // One tap of alignment class SA: pixels n < KS - SA read A, the others B.
    template <int SA>
    __device__ __forceinline__ void tap(const float *wt_tap)
    {
        MFL_SEG(5);
        float w[KS];
        const int wbase = (lane + KS - 1) * Cfg::RUNP + gi * KS;
#pragma unroll
        for (int n = 0; n < KS; ++n) w[n] = wt_tap[wbase - n * (Cfg::RUNP - 1)];
        if constexpr (BWD) {
            tap_scatter<SA>(w);
            return;
        }
        const uint32_t rec_new = request_record();
        const int voff = gvoff;
        MFL_SEG(0);
#ifndef MFL_ABL_NO_CORR
        // corrections first: their slot elements were requested a whole tap ago (right below, in the
        // previous tap), before the element loads of this tap, so waiting for them waits for nothing
        // that is still far away
        const uint32_t hdr = rec_word(rec_use, 2 * CH);
        if ((hdr >> 16) != 0) correct<SA>(hdr, w);
        MFL_SEG(1);
        if ((rec_word(rec_pre, 2 * CH) >> 16) != 0) load_slots(rec_pre);
        MFL_SEG(2);
#endif
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int n = 0; n < KS - SA; ++n) st.acc[i][n] = fmaf(w[n], v.A[i], st.acc[i][n]);
            v.A[i] = buf_load(rsrc, voff, rec_word(rec_pre, i));
        }
#pragma unroll
        for (int i = 0; i < CH; ++i) {
#pragma unroll
            for (int n = KS - SA; n < KS; ++n) st.acc[i][n] = fmaf(w[n], v.B[i], st.acc[i][n]);
            v.B[i] = buf_load(rsrc, voff, rec_word(rec_pre, CH + i));
        }
        MFL_SEG(3);
        rec_use = rec_pre;
        rec_pre = rec_req;
        rec_req = rec_new;
    }

    // Backward of one tap (the transpose of tap()): st.acc holds D = dy * x of the lane's pixels; what
    // the k pixels of a side send to their shared element is summed and stored.  An element of pad(x) belongs to
    // k lines (c', ky'), and within a line to the side A of one lane-step and the side B of another (mfl_diag.h):
    // (side, ky') selects one of 2k copies of the gradient planes in which this lane-step is the element's only
    // writer, so a plain store serves and the result does not depend on any order.  A flagged side splits the sum
    // between the shared and the extra element by the same rule the gather reads them; both are elements of the
    // same line and side.  (An earlier version added into one set of planes with 64-bit fixed-point atomics: 3.7 ms
    // at 21 labels, 512 x 512, 7 x 7, 8 images, bound by the rate of atomic operations.)
#ifdef MFL_ABL_NO_ATOMIC
    // timing experiment: the sums are formed, the stores replaced by one no launch performs
    template <bool EDGE>
    __device__ __forceinline__ void scatter_store(float *p, float v) const { if (v == 1234.5678f) *p = 1.f; }
#else
    // (lanes whose pixels are all outside the image hold D = 0: they store nothing, the copies were zeroed.  Where a
    // chain of an element's contributors leaves the pixel range, it continues in the neighbouring tap, whose alignment
    // class may differ: then two lanes send to the same side's copy.  Both are edge lanes, which add atomically; two
    // addends onto zero give the same sum in either order)
    // (a wave with an edge lane adds atomically in all its lanes - an atomic costs the same for six lanes as for 64,
    // and a store beside it in the other lanes costs more than it saves: 4.7 against 2.2 ms -; the two forms are
    // separate bodies chosen by a wave-uniform branch per tap.  The first and the last b-block are such waves.)
    template <bool EDGE>
    __device__ __forceinline__ void scatter_store(float *p, float v) const
    {
        if (v != 0.f) {
            if constexpr (EDGE) unsafeAtomicAdd(p, v);
            else *p = v;
        }
    }
#endif
    template <int SA>
    __device__ __forceinline__ void tap_scatter(const float (&w)[KS])
    {
        if (edge) tap_scatter_body<SA, true>(w);
        else tap_scatter_body<SA, false>(w);
    }
    template <int SA, bool EDGE>
    __device__ __forceinline__ void tap_scatter_body(const float (&w)[KS])
    {
        const uint32_t rec_new = request_record();
        const uint32_t hdr = rec_word(rec_use, 2 * CH);
        float *base = scatter_out + lane;
        constexpr int NA = KS - SA;
        auto threshold = [&](int x0) { return x0 + 63 > W - 1 ? W - x0 : (x0 < KS - 1 ? -x0 : 64); };
#pragma unroll
        for (int i = 0; i < CH; ++i) {
            if (ci * CH + i < C) {
                const uint32_t oA = rec_word(rec_use, i) >> 2, oB = rec_word(rec_use, CH + i) >> 2;
                const uint32_t kyw = rec_word(rec_use, diag_kyw_base(CH) + (i >> 2)) >> (8 * (i & 3));
                float *dstA = base + (size_t)(kyw & 15u) * copy_stride + oA;
                float *dstB = base + (size_t)(KS + ((kyw >> 4) & 15u)) * copy_stride + oB;
                if ((hdr >> i) & 1) {
                    const uint32_t x16 = (rec_word(rec_use, diag_xw_base(CH) + (i >> 1)) >> (16 * (i & 1))) & 0xffffu;
                    const int xA = x16 & 0x7fff;
                    const int xB = (x16 & 0x8000) ? (xA + 1 == W ? 0 : xA + 1) : xA;
                    const int pA = lane - threshold(xA), pB = lane - threshold(xB);
                    float sA = 0.f, eA = 0.f, sB = 0.f, eB = 0.f;
#pragma unroll
                    for (int n = 0; n < KS; ++n) {
                        const float t = w[n] * st.acc[i][n];
                        if (n < NA) { if (pA >= n) eA += t; else sA += t; }
                        else { if (pB >= n) eB += t; else sB += t; }
                    }
                    scatter_store<EDGE>(dstA, sA);
                    scatter_store<EDGE>(dstA + (gap4 >> 2), eA);
                    if constexpr (SA > 0) {
                        scatter_store<EDGE>(dstB, sB);
                        scatter_store<EDGE>(dstB + (gap4 >> 2), eB);
                    }
                } else {
                    float dA = 0.f, dB = 0.f;
#pragma unroll
                    for (int n = 0; n < KS; ++n) {
                        if (n < NA) dA = fmaf(w[n], st.acc[i][n], dA);
                        else dB = fmaf(w[n], st.acc[i][n], dB);
                    }
                    scatter_store<EDGE>(dstA, dA);
                    if constexpr (SA > 0) scatter_store<EDGE>(dstB, dB);
                }
            }
        }
        rec_use = rec_pre;
        rec_pre = rec_req;
        rec_req = rec_new;
    }

    // Pull the records of a group's taps into this XCD's L2 with one vector load (lane i reads the
    // first dword of tap i's record), so that the loads that fetch them a group later are L2 hits.
    // The value is folded into `touched`, which is only stored behind a condition no launch meets.
    uint32_t touched;
    __device__ __forceinline__ void touch_records(uint32_t gw)
    {
        const int kx = (gw >> 8) & 255, ky0 = (gw >> 16) & 255, nky = gw >> 24;
        const int i = lane < nky ? lane : nky - 1;
        touched |= table[(size_t)((ky0 + i) * KS + kx) * trec];
    }

    // all weight groups of alignment class SA, then the next class
    template <int SA>
    __device__ __forceinline__ void classes(int gidx)
    {
        const int bufsz = TB * Cfg::SLOTP;
        while (gidx < ngroups) {
            const uint32_t gw = groups[gidx];
            if ((int)(gw & 255) != SA) break;
            // The tap weights of the next group are produced alongside this group's taps: the image
            // loads of one weight tap are issued before a gather tap and turned into weights after it,
            // so their latency hides behind the tap.
            using WP = WeightPhase<CH, KS, NGW, NCS>;
            // (the last group produces the first group's weights once more into the free buffer, so
            // that every group runs the same branch-free loop)
            const uint32_t gn = groups[gidx + 1 < ngroups ? gidx + 1 : 0];
            const int nkyn = gn >> 24;
            if (gidx + 1 < ngroups) touch_records(gn);
            const int kxn = (gn >> 8) & 255, ky0n = (gn >> 16) & 255;
            const int dxn = kxn - p0;
            float *wnext = lds + ((gidx + 1) & 1) * bufsz;
            const float *wt = lds + (gidx & 1) * bufsz;
            const int nky = gw >> 24;
            const int ntap = nky > nkyn ? nky : nkyn;
            MFL_STAMP(3 + 3 * gidx);
            // Steady state: a tap of this group between the image loads and the weight stores of a
            // tap of the next group, with no branch in between (the wait counts before the stores then
            // skip over the tap's element loads instead of draining them).
            int tt = 0;
#if !defined(MFL_ABL_NO_WEIGHT) && !defined(MFL_ABL_NO_GATHER)
            const int ncommon = live ? (nky < nkyn ? nky : nkyn) : 0;
            for (; tt < ncommon; ++tt) {
                typename WP::Nb nb;
                WP::load(nb, st, irsrc, (ky0n + tt) * Wp + kxn);
                tap<SA>(wt + tt * Cfg::SLOTP);
                const int dyn = ky0n + tt - p0;
                WP::store(nb, st, (float)(dyn * dyn + dxn * dxn) * pw, cw, wnext + tt * Cfg::SLOTP);
            }
#endif
            for (; tt < ntap; ++tt) {
                typename WP::Nb nb;
#ifndef MFL_ABL_NO_WEIGHT
                if (tt < nkyn) WP::load(nb, st, irsrc, (ky0n + tt) * Wp + kxn);
#endif
#ifndef MFL_ABL_NO_GATHER
                if (live && tt < nky) tap<SA>(wt + tt * Cfg::SLOTP);
#endif
#ifndef MFL_ABL_NO_WEIGHT
                if (tt < nkyn) {
                    const int dyn = ky0n + tt - p0;
                    WP::store(nb, st, (float)(dyn * dyn + dxn * dxn) * pw, cw, wnext + tt * Cfg::SLOTP);
                }
#endif
            }
            MFL_STAMP(4 + 3 * gidx);
            __syncthreads();
            MFL_STAMP(5 + 3 * gidx);
            ++gidx;
        }
        if constexpr (SA + 1 < KS) classes<SA + 1>(gidx);
    }

    __device__ __forceinline__ void run()
    {
#ifdef MFL_STAMPS
        for (int i = 0; i < 6; ++i) seg[i] = 0;
        seg_t = __builtin_amdgcn_s_memtime();
#endif
        touched = 0;
        if (live) {
            pg = 0; ptt = 0;
            gw_cur = groups[0];
            rec_use = request_record();      // tap 0
            rec_pre = request_record();      // tap 1
            rec_req = request_record();      // tap 2
            if constexpr (!BWD) {
                const int voff = gvoff;
#pragma unroll
                for (int i = 0; i < CH; ++i) v.A[i] = buf_load(rsrc, voff, rec_word(rec_use, i));
#pragma unroll
                for (int i = 0; i < CH; ++i) v.B[i] = buf_load(rsrc, voff, rec_word(rec_use, CH + i));
                load_slots(rec_use);
            }
        }
        WeightPhase<CH, KS, NGW, NCS>::run(st, p0, Wp, irsrc, pw, cw, groups[0], lds);
        __syncthreads();
        MFL_STAMP(2);
        classes<0>(0);
    }
};

// MODE 0: fused mean-field update   MODE 1: y = x * window sums   MODE 2: y = window sums
// MODE 3: backward scatter: the 2k copies of the gradient planes (qp_out, zeroed by the caller, copy_stride floats
//         apart) receive the transpose of the window sums applied to
//         D = dy * x (dy passed as `unary`)
template <int CH, int KS, int NGW, int NCS, int MODE>
__global__ void __launch_bounds__(64 * NGW * NCS)
diag_kernel(DiagGeom g, int epi_rows, int tail_per_group, const float *__restrict__ xp, const F4 *__restrict__ ip4,
            const uint32_t *__restrict__ table, const uint32_t *__restrict__ groups, const float *__restrict__ pos_w,
            const float *__restrict__ col_w, const float *__restrict__ unary, const float *__restrict__ aw,
            const float *__restrict__ ab, const float *__restrict__ compat, const float *__restrict__ compat_plain,
            float *__restrict__ logq, float *__restrict__ qp_out, float *__restrict__ y_out, size_t copy_stride,
            unsigned long long *__restrict__ dbg)
{
    MFL_STAMP(0);
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    {
        // workgroups past the regular tiles: the pixels of a nearly empty last b-block
        const int regular = g.NBB * g.NGB * g.B;
        if ((int)blockIdx.x >= regular) {
            diag_tail_block<MODE>(g, lds, (int)blockIdx.x - regular, NGW * NCS, tail_per_group, xp, ip4, pos_w, col_w, unary,
                                  aw, ab, compat_plain, logq, qp_out, y_out);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave % NGW, ci = wave / NGW;
    // Workgroups are dealt round-robin over the 8 XCDs (each with its own L2); with the image index
    // fastest, image i of a batch of 8 stays on XCD i % 8.
    // (image-major order with one eighth of an image's tiles per XCD, which pays for 1024 x 2048 images in the
    // stream kernels, measured 5 % slower here: 8 x 22 MB of planes fit the memory-side cache either way)
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
        const int ibase = y * Wp + x;
        // (a slot without a pixel requests nothing: an offset past the descriptor's range returns 0 without a memory
        // request; its weights are never used)
        st.ioff[i] = valid ? ibase * 16 : (int)0x80000000u;
        st.lidx[i] = sigma < Cfg::NSLOT ? b_rel * Cfg::RUNP + a_rel : Cfg::SLOTP - 1;
        const F4 c = ip_b[ibase + ctr_off];
        st.cr[i] = c.x; st.cg[i] = c.y; st.cb[i] = c.z;
    }
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ip_b, 0, 0x7fffffff, 0x00020000);

    const int CBR = epi_rows;
    const int cstride = CBR * Cfg::RUNP;
    // pixel r of the chunk starting at b row cb0: LDS place, output pixel l and centre position in
    // the padded plane; false for the places of the tile that are no pixel of this workgroup
    auto place = [&](int cb0, int r, int &moff, int &l, int &cpos) {
        const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
        const int br = r / Cfg::RUN, a_rel = r - br * Cfg::RUN;
        const int b_rel = cb0 + br;
        const int n = a_rel % KS;
        const int owner = b_rel - (KS - 1) + n;
        const int aa = gb * Cfg::RUN + a_rel, bb = 64 * beta + b_rel - (KS - 1);
        const long long l64 = (long long)aa + (long long)bb * CK;
        const bool ok = br < rows && owner >= 0 && owner <= 63 && bb >= 0 && aa < CK && l64 < L;
        l = ok ? (int)l64 : 0;
        const int y = l / W, x = l - y * W;
        cpos = (y + g.p0) * Wp + x + g.p0;
        moff = ok ? br * Cfg::RUNP + a_rel : 0;
        return ok;
    };
    if constexpr (MODE == 3) {
        // backward: D = dy * x of the workgroup's pixels goes through LDS the other way round (threads
        // load whole pixels, coalesced; lanes pick up their diagonals), chunk by chunk
        const __amdgpu_buffer_rsrc_t rdy =
            __builtin_amdgcn_make_buffer_rsrc((void *)(unary + (size_t)img * C * L), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rxc = __builtin_amdgcn_make_buffer_rsrc((void *)(in_b + g.FS), 0, 0x7fffffff, 0x00020000);
        for (int cb0 = 0; cb0 < Cfg::NBR; cb0 += CBR) {
            const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
            const int cnt = rows * Cfg::RUN;
            for (int r = tid; r < cnt; r += Cfg::NT) {
                int moff, l, cpos;
                const bool ok = place(cb0, r, moff, l, cpos);
                const int br = r / Cfg::RUN, a_rel = r - br * Cfg::RUN;
                float *dst = lds + br * Cfg::RUNP + a_rel;
                // all loads first (channels past the launch's block re-read its last one), then the stores
                float dv[Cfg::CE], xv[Cfg::CE];
#pragma unroll
                for (int c = 0; c < Cfg::CE; ++c) {
                    const int cc = g.c0 + (c < g.Cv ? c : g.Cv - 1);
                    dv[c] = buf_load(rdy, l * 4, (uint32_t)(cc * L * 4));
                    xv[c] = buf_load(rxc, cpos * 4, (uint32_t)(cc * g.plane * 4));
                }
#pragma unroll
                for (int c = 0; c < Cfg::CE; ++c) dst[c * cstride] = (ok && c < g.Cv) ? dv[c] * xv[c] : 0.f;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < KS; ++n) {
                const int br = lane - n + KS - 1 - cb0;
                if (br >= 0 && br < CBR) {
                    const float *src = lds + (ci * CH * CBR + br) * Cfg::RUNP + gi * KS + n;
#pragma unroll
                    for (int i = 0; i < CH; ++i) st.acc[i][n] = src[i * cstride];
                }
            }
            __syncthreads();
        }
    }
    // A lane whose k pixels all lie past the last b value owns nothing (with many labels NB = L / CK is far below 64:
    // 36 at 150 labels, 512 x 512, 7 x 7): it requests no elements.  The kernel is bound by cache-line requests, and a
    // run of 42 lanes touches 2.3 lines where one of 64 touches 3.
#ifdef MFL_EXP_ALL_LANES_LOAD
    const int lane_voff = lane * 4;
#else
    const int lane_voff = (64 * beta + lane - (KS - 1) < g.NB) ? lane * 4 : (int)0x80000000u;
#endif
    // pixels (a0 + n, b - n) of this lane: the first has the largest index, the last the smallest.  The k contributors
    // of an element in one line are k pixels (CK - 1) apart, this lane's among them: where that chain may leave the
    // pixel range, the lane counts as an edge lane (Loop::scatter_store).
    bool edge = false;
    if constexpr (MODE == 3) {
        const long long l_first = (long long)gamma * KS + ((long long)64 * beta + lane) * CK;
        const long long l_last = l_first + (KS - 1) - (long long)(KS - 1) * CK;
        const long long span = (long long)(KS - 1) * (CK - 1);
        edge = __builtin_amdgcn_ballot_w64(l_last < span || l_first >= L - span) != 0;
    }
    Loop<CH, KS, NGW, NCS, MODE == 3> loop{qp_out + img_off, copy_stride, edge, g.Cv, ci, st, rsrc,
                                           table + diag_record_index(g, beta, live ? gamma : 0, 0, ci), groups, irsrc, lds,
                                           pw, cw, g.ngroups, g.TB, g.NCS * g.RECW, g.RECW, g.p0, Wp, W, g.gap * 4, gi,
                                           lane, lane_voff, live, dbg};
    MFL_STAMP(1);
    loop.run();
    if constexpr (MODE == 3) return;
    if (loop.touched == 0xfeedbeefu && g.B < 0) y_out[0] = 1.f;   // keeps the record prefetch loads alive
    MFL_STAMP(40);
#ifdef MFL_STAMPS
    if (dbg && blockIdx.x < 256 && threadIdx.x == 0)
        for (int i = 0; i < 6; ++i) dbg[blockIdx.x * 64 + 54 + i] = loop.seg[i];
#endif

    // ---- epilogue: accumulators -> LDS [channel][b_rel][a_rel], then one pixel per thread ----
    // accumulators of the b rows [cb0, cb0 + CBR) -> LDS.  The tile has room for all CH * NCS channel
    // rows, so the padded channels of the last split are written without a guard and never read.
    auto transpose = [&](int cb0) {
        if (live) {
#pragma unroll
            for (int n = 0; n < KS; ++n) {
                const int br = lane - n + KS - 1 - cb0;
                if (br >= 0 && br < CBR) {
                    float *dst = lds + (ci * CH * CBR + br) * Cfg::RUNP + gi * KS + n;
#pragma unroll
                    for (int i = 0; i < CH; ++i) dst[i * cstride] = st.acc[i][n];
                }
            }
        }
    };
    if constexpr (MODE == 1 || MODE == 2) {
        for (int cb0 = 0; cb0 < Cfg::NBR; cb0 += CBR) {
            transpose(cb0);
            __syncthreads();
            const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
            const int cnt = rows * Cfg::RUN;
#ifdef MFL_ABL_NO_EPI
            for (int r = tid; r < 1; r += Cfg::NT) {
#else
            for (int r = tid; r < cnt; r += Cfg::NT) {
#endif
                int moff, l, cpos;
                if (!place(cb0, r, moff, l, cpos)) continue;
                // (the launch's channels: all of them, or one block [c0, c0 + Cv) of a larger label set)
                const float *msg = lds + moff;
                const float *qc = in_b + g.FS + cpos + (size_t)g.c0 * g.plane;
                float *y_b = y_out + ((size_t)img * C + g.c0) * L + l;
                for (int c = 0; c < g.Cv; ++c)
                    y_b[(size_t)c * L] = MODE == 1 ? msg[c * cstride] * qc[(size_t)c * g.plane] : msg[c * cstride];
            }
            __syncthreads();
        }
    } else {
        // One pixel per thread and pass.  The pixel's unary and Q values do not depend on the gather,
        // so the passes are software pipelined: the loads of the next pass (possibly of the next
        // chunk) are issued before the arithmetic of the current one, the very first ones before
        // the accumulators go to LDS.  Channel planes are reached with scalar offsets (buffer
        // addressing): no per-channel address arithmetic in the vector unit.  Channels past C read
        // channel C-1 and are neutralised by value (unary -inf; their messages are finite and meet
        // zero columns of the padded label matrix), so that all loads issue back to back.
        constexpr int CE = Cfg::CE, CE2 = (CE + 1) & ~1;
        typedef float f2 __attribute__((ext_vector_type(2)));
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(in_b + g.FS), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t ru =
            __builtin_amdgcn_make_buffer_rsrc((void *)(unary + (size_t)img * C * L), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rqo =
            __builtin_amdgcn_make_buffer_rsrc((void *)(qp_out + img_off + g.FS), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rlq = __builtin_amdgcn_make_buffer_rsrc(
            (void *)((logq ? logq : qp_out) + (size_t)img * C * L), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t raw = __builtin_amdgcn_make_buffer_rsrc((void *)(aw ? aw : unary), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t rab = __builtin_amdgcn_make_buffer_rsrc((void *)(ab ? ab : unary), 0, 0x7fffffff, 0x00020000);
        const int plane4 = g.plane * 4, L4 = L * 4;
        struct Pixel {
            bool ok;
            int moff, l, cpos;
            float u[CE], qv[CE];
        };
        auto fetch = [&](int cb0, int it, Pixel &x) {
            x.ok = place(cb0, tid + it * Cfg::NT, x.moff, x.l, x.cpos);
#pragma unroll
            for (int c = 0; c < CE; ++c) {
                const int cc = c < C ? c : C - 1;
                x.qv[c] = buf_load(rq, x.cpos * 4, (uint32_t)(cc * plane4));
                x.u[c] = buf_load(ru, x.l * 4, (uint32_t)(cc * L4));
            }
        };
        // (wide channel capacities have no registers to spare for a second set of loaded values)
        constexpr bool kPrefetch = CE <= 24;
        Pixel nxt;
        if constexpr (kPrefetch) fetch(0, 0, nxt);
        for (int cb0 = 0; cb0 < Cfg::NBR; cb0 += CBR) {
            transpose(cb0);
            MFL_STAMP(42 + 4 * (cb0 != 0));
            __syncthreads();
            MFL_STAMP(43 + 4 * (cb0 != 0));
            const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
#ifdef MFL_ABL_NO_EPI
            const int nit = 0;
#else
            const int nit = (rows * Cfg::RUN + Cfg::NT - 1) / Cfg::NT;
#endif
            for (int it = 0; it < nit; ++it) {
                Pixel cur;
                if constexpr (kPrefetch) {
                    cur = nxt;
                    // the pass after this one (after the last one: this one again, loaded and dropped,
                    // which keeps the loop free of branches around loads)
                    const bool more = it + 1 < nit, chunk = cb0 + CBR < Cfg::NBR;
                    fetch(more ? cb0 : (chunk ? cb0 + CBR : cb0), more ? it + 1 : (chunk ? 0 : it), nxt);
                } else {
                    fetch(cb0, it, cur);
                }
                const float *msg = lds + cur.moff;
                const int l = cur.l, cpos = cur.cpos;
                float m[CE], u[CE];
#pragma unroll
                for (int c = 0; c < CE; ++c) {
                    const int cc = c < C ? c : C - 1;
                    m[c] = msg[cc * cstride];
                    u[c] = cur.u[c];
                }
                if (aw) {
#pragma unroll
                    for (int c = 0; c < CE; ++c) {
                        const int cc = c < C ? c : C - 1;
                        const float av = buf_load(raw, l * 4, (uint32_t)(cc * L4));
                        const float bv = buf_load(rab, l * 4, (uint32_t)(cc * L4));
                        m[c] = fmaf(av, m[c] * cur.qv[c], bv);
                    }
                } else {
#pragma unroll
                    for (int c = 0; c < CE; ++c) m[c] *= cur.qv[c];
                }
                float mu = -INFINITY;
#pragma unroll
                for (int c = 0; c < CE; ++c) {
                    // (channel splits leave at most 7 padded channels: only those need the test)
                    if (c >= CE - 7) u[c] = c < C ? u[c] : -INFINITY;
                    mu = fmaxf(mu, u[c]);
                }
                float su = 0.f;
#pragma unroll
                for (int c = 0; c < CE; ++c) su += __expf(u[c] - mu);
                const float lse_u = mu + __logf(su);
                // compat: the transposed, zero-padded CE2 x CE2 copy of the C x C matrix, so that a
                // pair of output channels is one packed FMA with a scalar register pair
                f2 pv[CE2 / 2];
                if (compat) {
                    const f2 *ct = reinterpret_cast<const f2 *>(compat);
#pragma unroll
                    for (int cp = 0; cp < CE2 / 2; ++cp) pv[cp] = f2{0.f, 0.f};
#pragma unroll
                    for (int c2 = 0; c2 < CE; ++c2)
#pragma unroll
                        for (int cp = 0; cp < CE2 / 2; ++cp)
                            pv[cp] = __builtin_elementwise_fma(ct[c2 * (CE2 / 2) + cp], f2{m[c2], m[c2]}, pv[cp]);
                } else {
#pragma unroll
                    for (int cp = 0; cp < CE2 / 2; ++cp) pv[cp] = f2{m[2 * cp], 2 * cp + 1 < CE ? m[2 * cp + 1] : 0.f};
                }
                float mv = -INFINITY;
#pragma unroll
                for (int c = 0; c < CE; ++c) {
                    u[c] = (u[c] - lse_u) - pv[c / 2][c & 1];
                    mv = fmaxf(mv, u[c]);
                }
                // Q_{t+1} = exp(u - mv) / sum: the exponentials of the sum are reused for the output
                float sv = 0.f;
#pragma unroll
                for (int c = 0; c < CE; ++c) {
                    u[c] -= mv;
                    m[c] = __expf(u[c]);
                    sv += m[c];
                }
                const float rsv = 1.f / sv;
#ifdef MFL_ABL_EPI_NOSTORE
                if (cur.ok && rsv == 1234.5f) {
#else
                if (cur.ok) {
#endif
#pragma unroll
                    for (int c = 0; c < CE; ++c)
                        if (c < C) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, m[c] * rsv), rqo, cpos * 4, c * plane4, 0);
                    if (logq) {
                        const float lsv = __logf(sv);
#pragma unroll
                        for (int c = 0; c < CE; ++c)
                            if (c < C) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, u[c] - lsv), rlq, l * 4, c * L4, 0);
                    }
                }
            }
            MFL_STAMP(44 + 4 * (cb0 != 0));
            __syncthreads();
            MFL_STAMP(45 + 4 * (cb0 != 0));
        }
    }
    MFL_STAMP(41);
}

template <int CH, int KS, int NGW, int NCS>
int launch_cfg(const DiagArgs &a, int mode, hipStream_t st)
{
    using Cfg = DiagCfg<CH, KS, NGW, NCS>;
    const DiagGeom &g = a.g;
    if (g.CH != CH || g.k != KS || g.NGW != NGW || g.NCS != NCS) return MFL_ERR_INVALID_ARGUMENT;
    // regular tiles, then the workgroups of the tail pixels (one wave per chunk of kTailPix pixels)
    const int tail_per_group = g.tail_lanes ? diag_tail_per_group(g) : 0;
    const long long tail_waves = (long long)g.B * g.NG * ((tail_per_group + kTailPix - 1) / kTailPix);
    const long long tail_blocks = (tail_waves + NGW * NCS - 1) / (NGW * NCS);
    const long long blocks = (long long)g.NBB * g.NGB * g.B + tail_blocks;
    if (blocks >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    if (((size_t)g.CK + (size_t)NGW * NCS * kTailPix * (g.K + 2 * g.C)) * 4 > a.lds_bytes) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)blocks);
    const size_t lds = a.lds_bytes;
    auto go = [&](auto kern) -> int {
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), lds, st, a.g, a.epi_rows, tail_per_group, a.xp, a.ip4, a.table, a.groups,
                           a.pos_w, a.col_w, a.unary, a.aw, a.ab, a.compat, a.compat_plain, a.logq, a.qp_out, a.y, a.copy_stride, a.dbg);
        return 0;
    };
    int rc;
    switch (mode) {
    case 0: rc = go(diag_kernel<CH, KS, NGW, NCS, 0>); break;
    case 1: rc = go(diag_kernel<CH, KS, NGW, NCS, 1>); break;
    case 2: rc = go(diag_kernel<CH, KS, NGW, NCS, 2>); break;
    case 3: rc = go(diag_kernel<CH, KS, NGW, NCS, 3>); break;
    default: return MFL_ERR_INVALID_ARGUMENT;
    }
    if (rc) return rc;
    return (int)hipGetLastError();
}

}  // namespace

namespace mfl {

#define MFL_CAT3_(a, b, c, d) a##b##c##d
#define MFL_CAT3(a, b, c, d) MFL_CAT3_(a, b, c, d)

int MFL_CAT3(launch_diag_ch, MFL_DIAG_CH, _k, MFL_DIAG_KS)(const DiagArgs &a, int mode, hipStream_t st)
{
    constexpr int CH = MFL_DIAG_CH, KS = MFL_DIAG_KS;
    const int ngw = a.g.NGW, ncs = a.g.NCS;
    if constexpr (CH == 8 || CH == 5) {
        if (ngw == 8 && ncs == 1) return launch_cfg<CH, KS, 8, 1>(a, mode, st);
        if (ngw == 4 && ncs == 2) return launch_cfg<CH, KS, 4, 2>(a, mode, st);
        if (ngw == 2 && ncs == 3) return launch_cfg<CH, KS, 2, 3>(a, mode, st);
        if constexpr (CH == 5) {
            if (ngw == 2 && ncs == 4) return launch_cfg<CH, KS, 2, 4>(a, mode, st);
        }
    } else {
        if (ngw == 4 && ncs == 2) return launch_cfg<CH, KS, 4, 2>(a, mode, st);
        if (ngw == 2 && ncs == 3) return launch_cfg<CH, KS, 2, 3>(a, mode, st);
#ifdef MFL_EXP_NGW
        if (ngw == MFL_EXP_NGW && ncs == 2) return launch_cfg<CH, KS, MFL_EXP_NGW, 2>(a, mode, st);
#endif
    }
    return MFL_ERR_INVALID_ARGUMENT;
}

}  // namespace mfl
