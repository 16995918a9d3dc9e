// Stream gather kernels of the local-window message-passing library (gfx950): the fast path for
// images at least 64 + k pixels wide.  Ownership map (a thread owns a diagonal of k output pixels that
// share one element of pad(x) per (channel, tap)): mfl_diag.h; tap stream and table records:
// mfl_ring.h.  One translation unit per (CH, KS): compile with -DMFL_RING_CH= -DMFL_RING_KS=.
//
// One workgroup = NGW consecutive groups x NCS channel splits (one wave each) of one b-block of one
// image.  In the tap loop the waves exchange nothing (no LDS traffic); one barrier per tap keeps them in
// step so that the cache lines they have in common (the weights of a group, overlapping element runs of
// neighbouring groups) are requested from the L2 once per workgroup.  The kernel is bound by the number of
// cache-line requests a CU can keep in flight, so lanes that own no pixels request nothing (their buffer
// offset is out of range), and a tap of alignment class 0 requests no side B.  Per stream tap a wave
//   - holds the k weights of each lane's pixels in registers, loaded a tap ahead from the weight table
//     (lane-major, 16 bytes per lane and load: full cache lines); the table is written once per call by
//     weights_kernel (the weights pd*pos_w + col_w*|I(l) - I(l + tap)|^2 do not change between the
//     iterations of a call): recomputing them per iteration cost more than half of all cache-line
//     requests of the step (one 12-byte image load per pixel and tap, in runs of k pixels);
//   - per channel: k FMAs on the shared elements (A, B) that were loaded during the previous tap,
//     then the loads of the next tap's (A, B) into the same registers (one coalesced dword per lane
//     each, scalar offsets straight from the table record: no per-lane address arithmetic).
// Epilogue: the accumulators go through LDS so that threads own whole pixels again (all channels,
// consecutive lanes = consecutive pixels) for the fused normalise step or the plain product.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mfl.h"
#include "mfl_ring_api.h"
#include "mfl_tail.h"

#if !defined(MFL_RING_CH) || !defined(MFL_RING_KS)
#error "compile with -DMFL_RING_CH=<channels per wave> -DMFL_RING_KS=<window side>"
#endif

using namespace mfl;

// cache policy of the streamed operands (weight table, unary, outputs): 2 = non-temporal.  Measured at 1024 x 2048,
// 11 x 11 (ms per step on 8 images): weight table 5.94 against 6.03 (the second reader of a weight line still finds it
// in L1, and the table, read once per step, displaces fewer element lines from the L2); unary 6.13; outputs 6.28
#ifndef MFL_NT_W
#define MFL_NT_W 2
#endif
#ifndef MFL_NT_U
#define MFL_NT_U 0
#endif
#ifndef MFL_NT_ST
#define MFL_NT_ST 0
#endif

// pixels per thread and epilogue chunk
#ifndef MFL_EPI_MULT
#define MFL_EPI_MULT 1
#endif
// waves per SIMD the fused step kernel is compiled for
#ifndef MFL_M0_WAVES
#define MFL_M0_WAVES 2
#endif
// ... and the plain window-sum kernels (modes 1, 2)
#ifndef MFL_M12_WAVES
#define MFL_M12_WAVES 4
#endif

namespace {

typedef float f3 __attribute__((ext_vector_type(3)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef const uint32_t __attribute__((address_space(4))) *ConstWords;   // scalar loads: the table is never written here

// Workgroups per image in the launch order: the tiles of an image padded to a multiple of 8 (one eighth per XCD).
__host__ __device__ inline int ring_blocks_per_image(const DiagGeom &g) { return ((g.NBB * g.NGB + 7) / 8) * 8; }

template <int CH, int KS, int NGW, int NCS>
struct RingCfg {
    static constexpr int NW = NGW * NCS;
    static constexpr int NT = 64 * NW;
    static constexpr int RUN = NGW * KS;
    static constexpr int RUNP = (RUN & 1) ? RUN + 2 : RUN + 1;   // pitch of the epilogue tile
    static constexpr int BL = 65 - KS;                            // lanes of a wave that own pixels
    static constexpr int NBR = BL + KS - 1;                       // b rows of a workgroup's tile (= 64)
    static constexpr int WQ = (KS + 3) / 4;                       // 16-byte pieces of a lane's k weights
    static constexpr int RECW = ((2 * CH + (CH + 1) / 2 + 1 + 3) / 4) * 4;
    static constexpr int CE = CH * NCS;
    static constexpr int CBR = (NT / RUN) * MFL_EPI_MULT;         // b rows per epilogue chunk: MFL_EPI_MULT pixels per thread
};

// MFL_ABL_* macros: timing experiments only; they remove one part of the kernel and give wrong results.

// Offsets of one tap (next tap's sides A and B) and the thresholds of one tap, held in scalar registers.
template <int CH>
struct Offsets {
    uint32_t a[CH], b[CH];
};
template <int CH>
struct Thresholds {
    uint32_t w[(CH + 1) / 2];
    uint32_t next;                          // bit s: step s of the next tap meets an image-row start
};

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

// The k multiply-adds of a step run as packed pairs (v_pk_fma_f32: two pixels per instruction; the tap loop is
// bound by instruction issue, one instruction per four cycles and SIMD with two waves resident).  A lane's k
// accumulators of a channel and its k weights are therefore held as k / 2 register pairs plus one single
// register, the shared elements (A, B) of a step as one pair.
template <int KS>
struct Weights {
    f2 p[KS / 2];
    float l;                                // pixel k - 1 (odd k)
    __device__ __forceinline__ void set(int n, float v)
    {
        if (n < 2 * (KS / 2)) p[n / 2][n & 1] = v;
        else l = v;
    }
    __device__ __forceinline__ float get(int n) const { return n < 2 * (KS / 2) ? p[n / 2][n & 1] : l; }
};
template <int CH, int KS>
struct Accumulators {
    f2 p[CH][KS / 2];
    float l[CH];
    __device__ __forceinline__ float get(int i, int n) const { return n < 2 * (KS / 2) ? p[i][n / 2][n & 1] : l[i]; }
};

// acc += w * e, per half: KIND 0 both pixels take e.x (side A), 1 both take e.y (side B), 2 the first takes
// e.x and the second e.y.  In place by construction (a three-address form makes the compiler copy the
// step's accumulators).
template <int KIND>
__device__ __forceinline__ void pk_fma(f2 &acc, f2 w, f2 e)
{
    if constexpr (KIND == 0) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(w), "v"(e));
    else if constexpr (KIND == 1) asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(w), "v"(e));
    else asm("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(e));
}

template <int CH, int KS, int NGW, int NCS>
struct Loop {
    using Cfg = RingCfg<CH, KS, NGW, NCS>;
    static constexpr int RECW = Cfg::RECW;

    Accumulators<CH, KS> &acc;
    __amdgpu_buffer_rsrc_t rsrc;            // FS floats before the pad(x) planes of the image
    __amdgpu_buffer_rsrc_t wrsrc;           // this wave's part of the weight table: [position][piece][lane][4]
    ConstWords recp;                        // record of the current stream position
    int rec_stride;                         // dwords between consecutive stream positions
    unsigned long long cols_kx, cols_sa;    // tap columns in stream order and their classes, 4 bits each
    int gap;
    int lane, bsrc;                         // bsrc: byte address of lane + gap for the lane permute
    int pos;                                // stream position
    Offsets<CH> offn;                       // offsets of the next tap (refills)
    Thresholds<CH> thr;                     // thresholds of the current tap
    f2 ab[CH];                              // shared elements (A, B) of the current tap, refilled in place
#ifdef MFL_STAMPS
    // diagnostic build only: cycles per tap segment, summed over the taps
    unsigned long long seg[6], seg_t;
#define MFL_SEG(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); seg[i] += t_ - seg_t; seg_t = t_; } while (0)
#else
#define MFL_SEG(i) do { } while (0)
#endif

    __device__ __forceinline__ void load_offsets(ConstWords p)
    {
#pragma unroll
        for (int c = 0; c < CH; ++c) { offn.a[c] = p[c]; offn.b[c] = p[CH + c]; }
    }
    __device__ __forceinline__ void load_thresholds(ConstWords p)
    {
#pragma unroll
        for (int j = 0; j < (CH + 1) / 2; ++j) thr.w[j] = p[2 * CH + j];
        thr.next = p[2 * CH + (CH + 1) / 2];
    }

    // One gather step: pixels n < KS - SA take side A, the others side B.
    template <int SA, int STEP>
    __device__ __forceinline__ void step(const Weights<KS> &w)
    {
        constexpr int NA = KS - SA, KP = KS / 2;
        const f2 e = ab[STEP];
#ifndef MFL_ABL_NO_FMA
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            if (2 * i + 1 < NA) pk_fma<0>(acc.p[STEP][i], w.p[i], e);
            else if (2 * i >= NA) pk_fma<1>(acc.p[STEP][i], w.p[i], e);
            else pk_fma<2>(acc.p[STEP][i], w.p[i], e);
        }
        if constexpr (KS & 1) acc.l[STEP] = fmaf(w.l, KS - 1 < NA ? e.x : e.y, acc.l[STEP]);
#endif
#ifndef MFL_ABL_NO_CORR
        const uint32_t t16 = (thr.w[STEP / 2] >> (16 * (STEP & 1))) & 0xffffu;
        if (t16 != 0x8080u) {
            // The 64-lane run meets the start of an image row: pixel n of a lane reads the element gap floats
            // further where lane - T >= n (mfl_diag.h, diag_side).  The FMAs above took the shared element for
            // every pixel; this adds w[n] * (extra - shared) where it applies.  The extra element of a lane
            // is the shared element of lane + gap: a lane permute (the lanes that own pixels end gap lanes
            // before the wave does).
            // (scalar copies first: __builtin_bit_cast applied to a vector element reads element 0 whichever is named)
            const float ea0 = e.x, eb0 = e.y;
            const float xa = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, ea0)));
            const float xb = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(bsrc, __builtin_bit_cast(int, eb0)));
            f2 d;
            d.x = xa - e.x;
            d.y = xb - e.y;
            const int pA = lane - ((int)(t16 & 255u) - 64), pB = lane - ((int)(t16 >> 8) - 64);
#pragma unroll
            for (int i = 0; i < KP; ++i) {
                f2 t;
                t.x = (2 * i < NA ? pA : pB) >= 2 * i ? w.p[i].x : 0.f;
                t.y = (2 * i + 1 < NA ? pA : pB) >= 2 * i + 1 ? w.p[i].y : 0.f;
                if (2 * i + 1 < NA) pk_fma<0>(acc.p[STEP][i], t, d);
                else if (2 * i >= NA) pk_fma<1>(acc.p[STEP][i], t, d);
                else pk_fma<2>(acc.p[STEP][i], t, d);
            }
            if constexpr (KS & 1) {
                const float t = (KS - 1 < NA ? pA : pB) >= KS - 1 ? w.l : 0.f;
                asm("v_fmac_f32 %0, %1, %2" : "+v"(acc.l[STEP]) : "v"(t), "v"(KS - 1 < NA ? d.x : d.y));
            }
        }
#endif
    }

    // NEXT_B: the next tap reads side B (a tap of class 0 does not: all its pixels take side A)
    template <int SA, int STEP, bool NEXT_B>
    __device__ __forceinline__ void steps(const Weights<KS> &w)
    {
#ifdef MFL_MID_BARRIER
        if constexpr (STEP > 0 && STEP % MFL_MID_BARRIER == 0) __builtin_amdgcn_s_barrier();   // timing experiment
#endif
        step<SA, STEP>(w);
#ifndef MFL_ABL_NO_GLOADS
        // the next tap's elements into the registers this step has just consumed; the lanes past BL own no
        // pixels and load along only where the next tap's step needs their elements as "extra" ones
        // (an offset out of the descriptor's range: the lane gets 0 and sends no request)
        const int gvoff = ((thr.next >> STEP) & 1) ? lane * 4 : (lane < Cfg::BL ? lane * 4 : (int)0x80000000u);
        ab[STEP].x = buf_load(rsrc, gvoff, offn.a[STEP]);
        if constexpr (NEXT_B) ab[STEP].y = buf_load(rsrc, gvoff, offn.b[STEP]);
#endif
        if constexpr (STEP + 1 < CH) steps<SA, STEP + 1, NEXT_B>(w);
    }

    // The k weights of the lane's pixels at stream position i (positions past the end repeat the last one).
    __device__ __forceinline__ void load_weights(Weights<KS> &w, int i) const
    {
        constexpr int K = KS * KS;
        const int ic = i < K ? i : K - 1;
#ifdef MFL_ABL_NO_WLOADS
        for (int n = 0; n < KS; ++n) w.set(n, (float)(ic + n));
        return;
#endif
        // (the lanes past BL own no pixels: their offset is out of the descriptor's range, which returns 0 without
        // a memory request, so a piece is 7 cache lines instead of 8; no branch, the wait counts stay exact)
        const int wvoff = lane < Cfg::BL ? lane * 16 : (int)0x80000000u;
#pragma unroll
        for (int q = 0; q < Cfg::WQ; ++q) {
            const f4 v = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, wvoff, (ic * Cfg::WQ + q) * 1024, MFL_NT_W));
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (4 * q + j < KS) w.set(4 * q + j, v[j]);
        }
    }

    // One tap of alignment class SA at stream position `pos` with the weights `w` (loaded a tap ago).
    template <int SA, bool NEXT_B = true>
    __device__ __forceinline__ void tap(Weights<KS> &w)
    {
        MFL_SEG(5);
        // requested now, used from the end of this tap on: the weights and thresholds of the next tap, the
        // element offsets of the tap after it (the stream's last record repeats itself)
        constexpr int K = KS * KS;
        Weights<KS> wn;
        // The waves of a workgroup start a tap together: the channel splits of a group request the same weights
        // and neighbouring groups overlapping element runs (side B of one is side A of the next, k floats
        // further), so whoever comes second finds the cache lines in the CU's L1 (32 KB: less than half a tap of
        // the CU's requests) instead of asking the L2 again.  Nothing is exchanged at the barrier; every wave of a
        // workgroup walks all K taps, so all of them arrive.
#ifndef MFL_NO_TAP_BARRIER
#ifdef MFL_TAP_BARRIER_EVERY
        if (pos % MFL_TAP_BARRIER_EVERY == 0)       // timing experiment
#endif
        __builtin_amdgcn_s_barrier();
#endif
        load_weights(wn, pos + 1);
#ifdef MFL_ABL_NO_RECORDS
        ConstWords rec1 = recp;
#else
        ConstWords rec1 = recp + rec_stride;
#endif
        Thresholds<CH> thr1;
        Offsets<CH> off2;
#pragma unroll
        for (int j = 0; j < (CH + 1) / 2; ++j) thr1.w[j] = rec1[2 * CH + j];
        thr1.next = rec1[2 * CH + (CH + 1) / 2];
        {
            ConstWords rec2 = pos + 2 <= K ? rec1 + rec_stride : rec1;
#pragma unroll
            for (int c = 0; c < CH; ++c) { off2.a[c] = rec2[c]; off2.b[c] = rec2[CH + c]; }
        }
        MFL_SEG(0);
        steps<SA, 0, NEXT_B>(w);
        MFL_SEG(1);
        recp = rec1;
        thr = thr1;
        offn = off2;
        w = wn;
        MFL_SEG(2);
        ++pos;
    }

    // All tap columns of alignment class SA (one unrolled loop body per class: the accumulators are
    // updated in place through one chain of loops, no two bodies meet in a common successor), then the
    // next class.
    template <int SA>
    __device__ __forceinline__ void classes(int col, Weights<KS> &w)
    {
        // (one flat loop per class: with a loop over the columns around a loop over the rows the compiler kept a
        // second copy of all accumulators across the inner loop, which doubled the kernel's register count)
        int ncols = 0;
        while (col + ncols < KS && (int)((cols_sa >> (4 * (col + ncols))) & 15) == SA) ++ncols;
        if constexpr (SA == 0) {
            // taps followed by another tap of class 0 request no side B; the last one requests it for the next class
            for (int i = ncols * KS - 1; i > 0; --i) tap<0, false>(w);
            if (ncols > 0) tap<0, true>(w);
        } else {
            for (int i = ncols * KS; i > 0; --i) tap<SA>(w);
        }
        if constexpr (SA + 1 < KS) classes<SA + 1>(col + ncols, w);
    }

    // Elements and weights of stream tap 0 requested, then the stream.
    __device__ __forceinline__ void run()
    {
        pos = 0;
        load_offsets(recp);
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            ab[c].x = buf_load(rsrc, lane * 4, offn.a[c]);
            ab[c].y = buf_load(rsrc, lane * 4, offn.b[c]);
        }
        load_thresholds(recp);
        load_offsets(recp + rec_stride);
        Weights<KS> w;
        load_weights(w, 0);
#ifdef MFL_STAMPS
        for (int i = 0; i < 6; ++i) seg[i] = 0;
        seg_t = __builtin_amdgcn_s_memtime();
#endif
        classes<0>(0, w);
    }
};

// Tap (ky, kx) at stream position i.
__device__ __forceinline__ void stream_tap(const RingGeom &r, int i, int &ky, int &kx)
{
    const int col = i / r.d.k;
    ky = i - col * r.d.k;
    kx = (int)((r.cols_kx >> (4 * col)) & 15);
}

// Weight table of a call: w[img][b-block][group][stream position][piece q][lane][4] with
// w = pd*pos_w + col_w*|I(l) - I(l + tap)|^2 (LocalConvolution.py:84 with the colour distance of
// AdvancedModel.py:547) for pixel n = 4q + j of the lane.
// One workgroup (eight waves) per (image, b-block, WG consecutive groups).  A lane of the gather owns a diagonal
// whose pixels are CK - 1 apart: loading along it touches one cache line per pixel.  Here the lanes of a wave
// walk the a direction instead: at one b value the WG * k pixels l = a + b*CK are CONSECUTIVE image pixels, one
// per lane, and the k - 1 lanes after them load the pixels to their right.  For one window row ky a lane loads
// ONE image pixel (row y + ky of its column); the k neighbours of that window row are the pixels of the next
// k lanes, reached by shifting the registers one lane per tap (v_mov_b32_dpp wave_shl:1): k loads per pixel
// instead of k * k.  A run that crosses the end of an image row (x0 + WG*k > W) takes the plain path, one
// load per lane and tap.  Each wave takes a share of the 64 + k - 1 b values that reach the block's 64 lanes.
// Pixel (a, b) belongs to lane (b - BL*beta) + a % k of group a / k: the weights go through LDS
// ([group][lane][n], two buffers, one barrier per tap) and leave lane-major, 1 KiB per store instruction.
template <int KS>
struct WeightsCfg {
    static constexpr int WG = (64 - (KS - 1)) / KS;        // groups per workgroup
    static constexpr int NPX = WG * KS;                     // their pixels along a
    static constexpr int NWV = 8;                           // waves per workgroup
};

__device__ __forceinline__ float shift_left_lane(float v)
{
    // lane i takes the value of lane i + 1 (the last lane takes 0)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

template <int KS>
__global__ void __launch_bounds__(64 * WeightsCfg<KS>::NWV) __attribute__((amdgpu_waves_per_eu(4, 4)))
weights_kernel(RingGeom rg, const F4 *__restrict__ ip4, const float *__restrict__ pos_w, const float *__restrict__ col_w,
               float *__restrict__ wtab)
{
    using WC = WeightsCfg<KS>;
    constexpr int WG = WC::WG, NPX = WC::NPX, NWV = WC::NWV, NT = 64 * NWV;
    constexpr int WQ = (KS + 3) / 4, WP = 4 * WQ + 1, K = KS * KS;
    constexpr int GST = 64 * WP + 17;                   // floats per group in LDS (the groups start on different banks)
    constexpr int ROWS = 64 + KS - 1;                   // b values that reach the 64 lanes of a b-block
    constexpr int RPW = (ROWS + NWV - 1) / NWV;         // ... per wave
    __shared__ float lds[2][WG * GST];
    const DiagGeom &g = rg.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ngq = (g.NG + WG - 1) / WG;
    const int gq = blockIdx.x % ngq;
    const int beta = (blockIdx.x / ngq) % g.NBB;
    const int img = blockIdx.x / (ngq * g.NBB);
    const F4 *__restrict__ ip_b = ip4 + (size_t)img * g.plane;
    const __amdgpu_buffer_rsrc_t irsrc = __builtin_amdgcn_make_buffer_rsrc((void *)ip_b, 0, 0x7fffffff, 0x00020000);
    const float pw = *pos_w, cw = *col_w;
    const int ctr_off = g.p0 * g.Wp + g.p0;
    // this lane's column: a = (first group)*k + lane, i.e. pixel n = lane % k of group gl = lane / k
    const int gl = lane / KS, n = lane - gl * KS;
    const long long a0 = (long long)gq * NPX;
    const bool col_ok = lane < NPX && a0 + lane < g.CK;
    int ioff[RPW], place[RPW];
    int rowoff[RPW];                                    // wave-uniform: the window origin of lane 0's pixel, in bytes
    uint32_t plain = 0;                                 // bit j: row j crosses the end of an image row (or is empty)
    float cr[RPW], cg[RPW], cb[RPW];
#pragma unroll
    for (int j = 0; j < RPW; ++j) {
        const int row = wave * RPW + j;
        const long long bb = (long long)g.BL * beta + row - (KS - 1);
        const long long l0 = a0 + bb * g.CK;            // pixel of lane 0
        const long long l = l0 + lane;
        const int owner = row - (KS - 1) + n;
        const bool ok = col_ok && row < ROWS && bb >= 0 && l < g.L && owner >= 0 && owner < g.BL;
        const int lc = ok ? (int)l : 0;
        const int yy = lc / g.W, xx = lc - yy * g.W;
        const int ibase = yy * g.Wp + xx;
        ioff[j] = ibase * 16;
        place[j] = ok ? gl * GST + owner * WP + n : -1;
        const F4 c = ip_b[ibase + ctr_off];
        cr[j] = c.x; cg[j] = c.y; cb[j] = c.z;
        const bool row_ok = row < ROWS && bb >= 0 && l0 < g.L;
        const int l0c = row_ok ? (int)l0 : 0;
        const int y0 = l0c / g.W, x0 = l0c - y0 * g.W;
        rowoff[j] = (y0 * g.Wp + x0) * 16;
        if (!row_ok || x0 + NPX > g.W) plain |= 1u << j;
    }
    // stream position of tap column kx, 4 bits each
    unsigned long long colpos = 0;
    for (int col = 0; col < KS; ++col) colpos |= (unsigned long long)col << (4 * (int)((rg.cols_kx >> (4 * col)) & 15));
    // zero both buffers once: places no thread writes (lanes that own no pixel, pixels past the image) stay 0
    for (int i = tid; i < 2 * WG * GST; i += NT) (&lds[0][0])[i] = 0.f;
    __syncthreads();
    const size_t unit0 = ((size_t)img * g.NBB + beta) * g.NG + (size_t)gq * WG;
    for (int ky = 0; ky < KS; ++ky) {
        // (no request ahead: with two workgroups per CU the other one works while these arrive, and the registers a
        // second set of pixels would take are what keeps the second workgroup off the CU)
        f3 cur[RPW];
#pragma unroll
        for (int j = 0; j < RPW; ++j)
            cur[j] = __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(irsrc, lane * 16, rowoff[j] + ky * g.Wp * 16, 0));
        const int dy = ky - g.p0;
        for (int kx = 0; kx < KS; ++kx) {
            const int it = ky * KS + kx;
            const int pos = (int)((colpos >> (4 * kx)) & 15) * KS + ky;
            const int dx = kx - g.p0;
            const float pdw = (float)(dy * dy + dx * dx) * pw;
            float *wt = lds[it & 1];
#pragma unroll
            for (int j = 0; j < RPW; ++j) {
                f3 nb = cur[j];
                if ((plain >> j) & 1)
                    nb = __builtin_bit_cast(f3, __builtin_amdgcn_raw_buffer_load_b96(irsrc, ioff[j], (ky * g.Wp + kx) * 16, 0));
                const float d0 = cr[j] - nb.x, d1 = cg[j] - nb.y, d2 = cb[j] - nb.z;
                const float cd = fmaf(d2, d2, fmaf(d1, d1, d0 * d0));
                if (place[j] >= 0) wt[place[j]] = fmaf(cd, cw, pdw);
                // the neighbour of the next tap column is the pixel one lane further
                cur[j].x = shift_left_lane(cur[j].x);
                cur[j].y = shift_left_lane(cur[j].y);
                cur[j].z = shift_left_lane(cur[j].z);
            }
            __syncthreads();
            // (two buffers: the next tap writes the other one; the barrier of that tap separates this tap's reads
            // from the writes of the tap after it)
            for (int i = tid; i < WG * WQ * 64; i += NT) {
                const int ol = i & 63, q = (i >> 6) % WQ, gg = i / (64 * WQ);
                // (the slots of the lanes past BL are never read but written along: whole 1 KiB pieces; skipping them
                // made this kernel 15 % slower)
                if (gq * WG + gg >= g.NG) continue;
                const float *src = wt + gg * GST + ol * WP + 4 * q;
                f4 v;
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) v[jj] = 4 * q + jj < KS ? src[jj] : 0.f;
                float *out = wtab + (unit0 + gg) * (size_t)K * (WQ * 256);
                // (written once, read an iteration later and far larger than any cache: non-temporal)
                __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(out + ((size_t)(pos * WQ + q) * 64 + ol) * 4));
            }
        }
    }
}

// MODE 0: fused mean-field update   MODE 1: y = x * window sums   MODE 2: y = window sums
template <int CH, int KS, int NGW, int NCS, int MODE>
__global__ void __launch_bounds__(64 * NGW * NCS) __attribute__((amdgpu_waves_per_eu(MODE == 0 || CH >= 10 ? MFL_M0_WAVES : MFL_M12_WAVES)))
ring_kernel(RingGeom rg, int lds_floats, int tail_per_group, const float *__restrict__ xp, const F4 *__restrict__ ip4,
            const uint32_t *__restrict__ table, const float *__restrict__ wtab, const float *__restrict__ pos_w, const float *__restrict__ col_w,
            const float *__restrict__ unary, const float *__restrict__ aw, const float *__restrict__ ab,
            const float *__restrict__ compat, const float *__restrict__ compat_plain, float *__restrict__ logq,
            float *__restrict__ qp_out, float *__restrict__ y_out)
{
    using Cfg = RingCfg<CH, KS, NGW, NCS>;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const DiagGeom &g = rg.d;
    {
        // workgroups past the regular tiles: the pixels of a nearly empty last b-block
        const int regular = ring_blocks_per_image(g) * g.B;
        if ((int)blockIdx.x >= regular) {
            diag_tail_block<MODE>(g, lds, (int)blockIdx.x - regular, Cfg::NW, tail_per_group, xp, ip4, pos_w, col_w, unary, aw, ab,
                                  compat_plain, logq, qp_out, y_out);
            return;
        }
    }
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gi = wave % NGW, ci = wave / NGW;
    // Image-major launch order: the workgroups of one image run together, so that the image's planes (far
    // larger than an L2 at 1024 x 2048) stay in the memory-side cache while their ~k re-reads happen, instead of
    // all images of the batch competing for it.  Within an image, workgroups are dealt round-robin over the 8
    // XCDs by the hardware; XCD x takes the x-th contiguous eighth of the image's tiles, so that neighbouring
    // tiles (which share cache lines) meet in one L2.
    const int per_image = ring_blocks_per_image(g);
    const int img = blockIdx.x / per_image;
    const int rblk = blockIdx.x - img * per_image;
    const int tile = (rblk & 7) * (per_image >> 3) + (rblk >> 3);
    if (tile >= g.NBB * g.NGB) return;                // (the padding of the last eighth)
    // Tile order within an image: b-blocks in patches of kPatch, the group blocks of a patch in turn, its b-blocks
    // fastest.  Workgroups dispatched one after the other run a few taps apart, so their common cache lines meet in
    // the XCD's L2: neighbouring group blocks overlap in whole element runs, neighbouring b-blocks in the line their
    // runs straddle (b-blocks slowest: 4 % slower at 1024 x 2048, 11 x 11; b-blocks fastest: 2 % slower still).
    constexpr int kPatch = 2;
    int gb, beta;
    {
        const int full = (g.NBB / kPatch) * kPatch;            // b-blocks in complete patches
        const int per_patch = kPatch * g.NGB;
        if (tile < (full / kPatch) * per_patch) {
            const int patch = tile / per_patch, r = tile - patch * per_patch;
            gb = r / kPatch;
            beta = patch * kPatch + r % kPatch;
        } else {
            const int r = tile - (full / kPatch) * per_patch, rest = g.NBB - full;
            gb = r / rest;
            beta = full + r % rest;
        }
    }
    const int gamma = gb * NGW + gi;
    const bool live = gamma < g.NG;
    const int C = g.C, CK = g.CK, L = g.L, W = g.W, Wp = g.Wp;

    const size_t img_off = (size_t)img * C * g.plane;
    const float *in_b = xp + img_off;                 // FS floats before the image's padded planes
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)in_b, 0, 0x7fffffff, 0x00020000);
    Accumulators<CH, KS> acc;
#pragma unroll
    for (int i = 0; i < CH; ++i) {
#pragma unroll
        for (int n = 0; n < KS / 2; ++n) acc.p[i][n] = f2{0.f, 0.f};
        acc.l[i] = 0.f;
    }

    const int gamma_t = live ? gamma : g.NG - 1;      // a wave without a group works on the last group (results dropped)
    const size_t wunit = ((size_t)img * g.NBB + beta) * g.NG + gamma_t;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        (void *)(wtab + wunit * (size_t)(KS * KS) * (Cfg::WQ * 256)), 0, 0x7fffffff, 0x00020000);
#ifdef MFL_EXP_PAIR
    // timing experiment (wrong results): the second half of a workgroup's groups walks the records of the first half,
    // i.e. requests exactly the element lines its partner requests - what a workgroup made of units that read the
    // same rows of pad(x) at the same time would see; the weights stay each unit's own
    const int gamma_rec = live ? gb * NGW + (gi % (NGW / 2)) : g.NG - 1;
#else
    const int gamma_rec = gamma_t;
#endif
    Loop<CH, KS, NGW, NCS> loop{acc, rsrc, wrsrc,
                                (ConstWords)(uintptr_t)(table + ring_record_index(rg, beta, gamma_rec, 0, ci)),
                                g.NCS * g.RECW, rg.cols_kx, rg.cols_sa, g.gap, lane, ((lane + g.gap) & 63) * 4, 0, {}, {}, {}};
#ifdef MFL_STAMPS
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    loop.run();
#ifdef MFL_STAMPS
    unsigned long long *dbg = reinterpret_cast<unsigned long long *>(y_out);
    const unsigned long long t_loop = __builtin_amdgcn_s_memtime();
#endif
    __syncthreads();

    // ---- epilogue: accumulators -> LDS [channel][b_rel][a_rel], then one pixel per thread ----
    constexpr int CBR = Cfg::CBR;                    // b rows per chunk
    static_assert(CBR >= 1, "a workgroup's threads cover at least one b row of its tile");
    const int cstride = CBR * Cfg::RUNP;
    // pixel r of the chunk starting at b row cb0: LDS place, output pixel l and centre position in
    // the padded plane; false for the places of the tile that are no pixel of this workgroup
    auto place = [&](int cb0, int r, int &moff, int &l, int &cpos) {
        const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
        const int br = r / Cfg::RUN, a_rel = r - br * Cfg::RUN;
        const int b_rel = cb0 + br;
        const int n = a_rel % KS;
        const int owner = b_rel - (KS - 1) + n;
        const int aa = gb * Cfg::RUN + a_rel, bb = Cfg::BL * beta + b_rel - (KS - 1);
        const long long l64 = (long long)aa + (long long)bb * CK;
        const bool ok = br < rows && owner >= 0 && owner < Cfg::BL && bb >= 0 && aa < CK && l64 < L;
        l = ok ? (int)l64 : 0;
        const int yy = l / W, xx = l - yy * W;
        cpos = (yy + g.p0) * Wp + xx + g.p0;
        moff = ok ? br * Cfg::RUNP + a_rel : 0;
        return ok;
    };
    // accumulators of the b rows [cb0, cb0 + CBR) -> LDS.  The tile has room for all CH * NCS channel
    // rows, so the padded channels of the last split are written without a guard and never read.
    auto transpose = [&](int cb0) {
        if (live && lane < Cfg::BL) {
#pragma unroll
            for (int n = 0; n < KS; ++n) {
                const int br = lane - n + KS - 1 - cb0;
                if (br >= 0 && br < CBR) {
                    float *dst = lds + (ci * CH * CBR + br) * Cfg::RUNP + gi * KS + n;
#pragma unroll
                    for (int i = 0; i < CH; ++i) dst[i * cstride] = acc.get(i, n);
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
            for (int r = tid; r < cnt; r += Cfg::NT) {
                int moff, l, cpos;
                if (!place(cb0, r, moff, l, cpos)) continue;
                const float *msg = lds + moff;
                const float *qc = in_b + g.FS + cpos;
                float *y_b = y_out + (size_t)img * C * L + l;
                for (int c = 0; c < C; ++c)
                    y_b[(size_t)c * L] = MODE == 1 ? msg[c * cstride] * qc[(size_t)c * g.plane] : msg[c * cstride];
            }
            __syncthreads();
        }
    } else {
        // Two lanes per pixel: lane (p, h) = (lane & 31, lane >> 5) of a wave works on pixel p of the
        // wave's 32 pixels of a pass.  The label-compatibility product pv[c] = sum_c2 compat[c][c2] * m[c2]
        // is one chain of v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate) per pass:
        //   A (rows = output channels)  lane holds ct[2j + h][lane & 31] for k-step j: the 32 x 32
        //                               transposed, zero-padded matrix, in registers for the whole epilogue
        //   B (columns = pixels)        lane holds m[p][2j + h] = msg * Q (input channels 2j + h)
        //   D                           register v of lane (p, h) = pv[p][8 (v / 4) + 4 h + v % 4]
        // so a lane normalises the output channels {8 g + 4 h + r} of its pixel and the two lanes of a
        // pixel exchange the maxima and sums of the two softmaxes.  All loads and stores are buffer
        // accesses whose range check (offset >= C planes) drops the channels past C: loads return 0.
        constexpr int CE = Cfg::CE;
        constexpr int KJ = CE >= 32 ? 16 : (CE + 1) / 2;            // k-steps (input channels 2j, 2j + 1)
        constexpr int NV = CE >= 32 ? 16 : 4 * ((CE + 7) / 8);      // D registers in use
        typedef float f16v __attribute__((ext_vector_type(16)));
        const int p = lane & 31, h = lane >> 5;
        const int plane4 = g.plane * 4, L4 = L * 4;
        const uint32_t oob = 0x80000000u;
        const __amdgpu_buffer_rsrc_t rq = __builtin_amdgcn_make_buffer_rsrc((void *)(in_b + g.FS), 0, C * plane4, 0x00020000);
        const __amdgpu_buffer_rsrc_t ru =
            __builtin_amdgcn_make_buffer_rsrc((void *)(unary + (size_t)img * C * L), 0, C * L4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rqo =
            __builtin_amdgcn_make_buffer_rsrc((void *)(qp_out + img_off + g.FS), 0, C * plane4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rlq =
            __builtin_amdgcn_make_buffer_rsrc((void *)((logq ? logq : qp_out) + (size_t)img * C * L), 0, C * L4, 0x00020000);
        const __amdgpu_buffer_rsrc_t raw = __builtin_amdgcn_make_buffer_rsrc((void *)(aw ? aw : unary), 0, C * L4, 0x00020000);
        const __amdgpu_buffer_rsrc_t rab = __builtin_amdgcn_make_buffer_rsrc((void *)(ab ? ab : unary), 0, C * L4, 0x00020000);
        auto bload = [](__amdgpu_buffer_rsrc_t r, uint32_t voff) {
#ifdef MFL_ABL_EPI_NOLOAD
            return __builtin_bit_cast(float, voff & 0x3fffffffu);
#endif
            return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, 0, 0));
        };
        auto other_half = [&](float v) {
            return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
        };
        float am[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) am[j] = compat[(2 * j + h) * 32 + p];
        // output channels of this lane that exist: bit v
        uint32_t vmask = 0;
#pragma unroll
        for (int v = 0; v < NV; ++v) vmask |= (uint32_t)(8 * (v / 4) + 4 * h + (v % 4) < C) << v;
#ifdef MFL_STAMPS
        unsigned long long es[24];
        int esn = 0;
#define MFL_ES() do { if (esn < 24) es[esn++] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define MFL_ES() do { } while (0)
#endif
        for (int cb0 = 0; cb0 < Cfg::NBR; cb0 += CBR) {
            MFL_ES();
            transpose(cb0);
            MFL_ES();
            __syncthreads();
            MFL_ES();
            const int rows = (Cfg::NBR - cb0 < CBR) ? Cfg::NBR - cb0 : CBR;
#ifdef MFL_ABL_NO_EPI
            const int nit = 0;
#else
            const int nit = (rows * Cfg::RUN + 32 * Cfg::NW - 1) / (32 * Cfg::NW);
#endif
            for (int it = 0; it < nit; ++it) {
                int moff, l, cpos;
                const bool ok = place(cb0, (it * Cfg::NW + wave) * 32 + p, moff, l, cpos);
#if defined(MFL_ABL_EPI_UCOAL) || defined(MFL_ABL_EPI_QCOAL)
                // timing experiments (wrong results): the epilogue's unary / Q accesses as if a wave's 32 pixels were
                // consecutive and 128-byte aligned in memory
                const uint32_t fake = ((uint32_t)(blockIdx.x % 4096) * 2048u + (uint32_t)cb0 * 32u + (uint32_t)((it * Cfg::NW + wave) * 32 + p));
#endif
#ifdef MFL_ABL_EPI_QCOAL
                const uint32_t cposq = fake % (uint32_t)(g.plane - 64);
#else
                const uint32_t cposq = (uint32_t)cpos;
#endif
#ifdef MFL_ABL_EPI_UCOAL
                const uint32_t lu = fake % (uint32_t)L;
#else
                const uint32_t lu = (uint32_t)l;
#endif
                const uint32_t vq = (uint32_t)(cposq * 4 + h * plane4), vu = (uint32_t)(lu * 4 + 4 * h * L4);
                float m[KJ], u[NV];
#pragma unroll
                for (int j = 0; j < KJ; ++j) m[j] = bload(rq, vq + (uint32_t)(2 * j * plane4));
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    u[v] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(ru, (int)(vu + (uint32_t)((8 * (v / 4) + (v % 4)) * L4)), 0, MFL_NT_U));
                {
                    const float *msg = lds + moff;
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        // (the row of a channel past the tile: any row, its Q factor is 0)
                        const int row = 2 * j + 1 < CE ? 2 * j + h : (2 * j < CE ? 2 * j : 0);
                        m[j] *= msg[row * cstride];
                    }
                }
                if (aw) {
                    const uint32_t va = (uint32_t)(l * 4 + h * L4);
#pragma unroll
                    for (int j = 0; j < KJ; ++j)
                        m[j] = fmaf(bload(raw, va + (uint32_t)(2 * j * L4)), m[j], bload(rab, va + (uint32_t)(2 * j * L4)));
                }
                f16v d;
#pragma unroll
                for (int v = 0; v < 16; ++v) d[v] = 0.f;
#pragma unroll
                for (int j = 0; j < KJ; ++j) d = __builtin_amdgcn_mfma_f32_32x32x2f32(am[j], m[j], d, 0, 0, 0);
                float mu = -INFINITY;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    u[v] = (vmask >> v) & 1 ? u[v] : -INFINITY;
                    mu = fmaxf(mu, u[v]);
                }
                mu = fmaxf(mu, other_half(mu));
                float su = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) su += __expf(u[v] - mu);
                su += other_half(su);
                const float lse_u = mu + __logf(su);
                float mv = -INFINITY;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    u[v] = (u[v] - lse_u) - d[v];
                    mv = fmaxf(mv, u[v]);
                }
                mv = fmaxf(mv, other_half(mv));
                // Q_{t+1} = exp(u - mv) / sum (the exponentials are evaluated again for the output rather than
                // held in registers across the exchange)
                float sv = 0.f;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    u[v] -= mv;
                    sv += __expf(u[v]);
                }
                sv += other_half(sv);
                const float rsv = 1.f / sv;
#ifdef MFL_ABL_EPI_NOSTORE
                const uint32_t sq = rsv == 12345.f ? (uint32_t)(cpos * 4 + 4 * h * plane4) : oob;
#else
                const uint32_t sq = ok ? (uint32_t)(cposq * 4 + 4 * h * plane4) : oob;
#endif
#pragma unroll
                for (int v = 0; v < NV; ++v)
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, __expf(u[v]) * rsv), rqo,
                                                          (int)(sq + (uint32_t)((8 * (v / 4) + (v % 4)) * plane4)), 0, MFL_NT_ST);
                if (logq) {
                    const float lsv = __logf(sv);
                    const uint32_t sl = ok ? vu : oob;
#pragma unroll
                    for (int v = 0; v < NV; ++v)
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, u[v] - lsv), rlq,
                                                              (int)(sl + (uint32_t)((8 * (v / 4) + (v % 4)) * L4)), 0, 0);
                }
            }
            MFL_ES();
            __syncthreads();
        }
        MFL_ES();
#ifdef MFL_STAMPS
        if (dbg && blockIdx.x < 512 && lane == 0) {
            unsigned long long *dd = dbg + ((size_t)blockIdx.x * Cfg::NW + wave) * 48;
            for (int i = 0; i < 24; ++i) dd[16 + i] = i < esn ? es[i] - t_loop : 0;
        }
#endif
    }
#ifdef MFL_STAMPS
    if (MODE == 0 && dbg && blockIdx.x < 512 && lane == 0) {
        unsigned long long *d = dbg + ((size_t)blockIdx.x * Cfg::NW + wave) * 48;
        for (int i = 0; i < 6; ++i) d[i] = loop.seg[i];
        d[6] = t_loop - t_begin;
        d[7] = __builtin_amdgcn_s_memtime() - t_loop;
        d[8] = t_begin;
    }
#endif
}

template <int CH, int KS, int NGW, int NCS>
size_t lds_bytes_cfg(const RingGeom &r)
{
    using Cfg = RingCfg<CH, KS, NGW, NCS>;
    const DiagGeom &g = r.d;
    size_t need = 0;
    // the epilogue tile: as many b rows of all channels as give every thread one pixel per chunk
    const size_t epi = (size_t)Cfg::CE * Cfg::RUNP * Cfg::CBR * 4;
    if (epi > need) need = epi;
    const size_t tail = ((size_t)g.CK + (size_t)Cfg::NW * kTailPix * (g.K + 2 * g.C)) * 4;
    if (g.tail_lanes && tail > need) need = tail;
#ifdef MFL_EXP_LDS_MIN
    if (need < (size_t)MFL_EXP_LDS_MIN) need = MFL_EXP_LDS_MIN;    // timing experiment: fewer workgroups per CU
#endif
    return (need + 255) & ~(size_t)255;
}

template <int CH, int KS, int NGW, int NCS>
int launch_cfg(const RingArgs &a, int mode, hipStream_t st)
{
    using Cfg = RingCfg<CH, KS, NGW, NCS>;
    const DiagGeom &g = a.r.d;
    if (g.CH != CH || g.k != KS || g.NGW != NGW || g.NCS != NCS || g.RECW != Cfg::RECW)
        return MFL_ERR_INVALID_ARGUMENT;
    // regular tiles, then the workgroups of the tail pixels (one wave per chunk of kTailPix pixels)
    const int tail_per_group = g.tail_lanes ? diag_tail_per_group(g) : 0;
    const long long tail_waves = (long long)g.B * g.NG * ((tail_per_group + kTailPix - 1) / kTailPix);
#ifdef MFL_ABL_NO_TAIL
    const long long tail_blocks = 0;                  // timing experiment: the tail pixels are not computed
#else
    const long long tail_blocks = (tail_waves + Cfg::NW - 1) / Cfg::NW;
#endif
    const long long blocks = (long long)ring_blocks_per_image(g) * g.B + tail_blocks;
    if (blocks >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const size_t lds = lds_bytes_cfg<CH, KS, NGW, NCS>(a.r);
    if (lds > 160 * 1024) return MFL_ERR_INVALID_ARGUMENT;
    const dim3 grid((unsigned)blocks);
    auto go = [&](auto kern) -> int {
        // the attribute belongs to the current device's copy of the kernel: set before every launch (a process may
        // drive several devices; the call is cheap)
        const hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
        hipLaunchKernelGGL(kern, grid, dim3(Cfg::NT), lds, st, a.r, (int)(lds / 4), tail_per_group, a.xp, a.ip4, a.table, a.wtab, a.pos_w,
                           a.col_w, a.unary, a.aw, a.ab, a.compat, a.compat_plain, a.logq, a.qp_out, a.y);
        return 0;
    };
    if (mode == 3) {
        // the weight table of the call
        constexpr int WG = WeightsCfg<KS>::WG;
        const long long wgs = (long long)g.B * g.NBB * ((g.NG + WG - 1) / WG);
        if (wgs >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
        hipLaunchKernelGGL(weights_kernel<KS>, dim3((unsigned)wgs), dim3(64 * WeightsCfg<KS>::NWV), 0, st, a.r, a.ip4, a.pos_w, a.col_w, a.wtab_out);
        return (int)hipGetLastError();
    }
    int rc;
    switch (mode) {
    case 0: rc = go(ring_kernel<CH, KS, NGW, NCS, 0>); break;
    case 1: rc = go(ring_kernel<CH, KS, NGW, NCS, 1>); break;
    case 2: rc = go(ring_kernel<CH, KS, NGW, NCS, 2>); break;
    default: return MFL_ERR_INVALID_ARGUMENT;
    }
    if (rc) return rc;
    return (int)hipGetLastError();
}

}  // namespace

namespace mfl {

#define MFL_CAT4_(a, b, c, d) a##b##c##d
#define MFL_CAT4(a, b, c, d) MFL_CAT4_(a, b, c, d)

int MFL_CAT4(launch_ring_ch, MFL_RING_CH, _k, MFL_RING_KS)(const RingArgs &a, int mode, hipStream_t st)
{
    constexpr int CH = MFL_RING_CH, KS = MFL_RING_KS;
    const int ngw = a.r.d.NGW, ncs = a.r.d.NCS;
    if (ngw == 2 && ncs == 2) return launch_cfg<CH, KS, 2, 2>(a, mode, st);
    if (ngw == 2 && ncs == 3) return launch_cfg<CH, KS, 2, 3>(a, mode, st);
    if (ngw == 1 && ncs == 4) return launch_cfg<CH, KS, 1, 4>(a, mode, st);
#ifdef MFL_EXPERIMENTS
    if (ngw == 1 && ncs == 3) return launch_cfg<CH, KS, 1, 3>(a, mode, st);
    if (ngw == 4 && ncs == 2) return launch_cfg<CH, KS, 4, 2>(a, mode, st);
#endif
    return MFL_ERR_INVALID_ARGUMENT;
}

}  // namespace mfl
