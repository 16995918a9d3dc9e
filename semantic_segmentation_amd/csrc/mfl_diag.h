// Geometry, ownership map and address tables of the "diagonal" gather kernels
// (mfl_diag.hip): the fast path of the local-window message passing for images at
// least 64 + k pixels wide.  Plain C++ shared by the device code and by the host
// emulation in the CPU test-suite.
//
// Arithmetic restated (not code) from Conv2dLocal.forward, LocalConvolution.py:59-90.
//
// ---- the index map, seen along its diagonals ---------------------------------
// With K = k*k, L = H*W, CK = C*K, output (c, l), tap t reads V[g], g = (c*K + t)*L + l,
// where V[l'*CK + c'*K + ky'*k + kx'] = pad(x)[c'][y' + ky'][x' + kx']  (mfl_core.h).
// Write an output pixel as l = a + b*CK (a < CK) and g = G + n + b*CK with
// G = (c*K + t)*L + a0 = l0'*CK + r0.  Adding CK to g moves to the next source pixel
// (address + 1), adding 1 moves to the next kx' (address + 1), so the pixels
//     (a0 + n, b - n),  n = 0 .. k-1
// all read the SAME element of pad(x) as long as r0 % k + n < k; the others read one
// common element of the next (c', ky') line.  A thread therefore owns such a diagonal
// of k output pixels: per (channel, tap) it needs two loaded values (A, B) for k FMAs,
// and the 64 lanes of a wave, which own consecutive b, load 64 consecutive floats.
// s = r0 % k = (t*L) % k depends on the tap column kx only, so which pixels use A or B
// is known per tap column and the kernels are specialised on it.
//
// The sharing breaks where the run of source pixels meets the start of an image row
// (the zero padding of pad(x) separates rows by k-1 floats): such (channel, tap) steps
// are flagged in the table and take a per-lane corrected path.
//
// ---- who computes what ---------------------------------------------------------
// group      gamma = a0 / k           (C*k groups)
// b-block    beta: lanes own b = BL*beta + lane - n (BL = 64 here)
// channels are split over NCS waves of CH channels each; a workgroup is NGW consecutive
// groups x NCS channel splits and shares one table of tap weights in LDS.
#ifndef MFL_DIAG_H
#define MFL_DIAG_H

#include <stddef.h>
#include <stdint.h>

#include "mfl_core.h"

namespace mfl {

struct DiagGeom {
    int B, C, H, W, k, K, p0;
    int c0, Cb, Cv;           // output channels of a launch: [c0, c0 + Cv) of a block of width Cb (the split is made for
                              // Cb); c0 = 0, Cb = Cv = C unless the channels are computed block by block (more labels
                              // than one workgroup's waves hold)
    int L, CK;
    int Hp, Wp, plane, gap;   // zero-padded plane; gap = Wp - W = k - 1
    int NB;                   // number of b values: ceil(L / CK)
    int NG;                   // groups: C*k
    int BL;                   // b values per block: 64 (diagonal kernels) or 65 - k (stream kernels: the last k - 1
                              // lanes of a wave only load elements the other lanes need at image-row starts)
    int NBB;                  // b-blocks the kernels run
    int tail_lanes;           // lanes of one more, nearly empty b-block whose few pixels go to the tail
                              // kernel instead (0: none)
    int NGW, NCS, CH;         // workgroup = NGW groups x NCS channel splits; CH channels per wave
    int NGB;                  // group blocks: ceil(NG / NGW)
    int RUN, RUNP;            // pixels of a workgroup along a (NGW*k) and the odd LDS pitch
    int NBR;                  // b values a workgroup touches: 64 + k - 1
    int Lk;                   // L % k
    int FS;                   // floats of zeroed slack in front of the padded planes
    int RECW;                 // dwords per table record
    int TB;                   // taps per weight group
    int ngroups;              // weight groups per iteration
};

// Floats of zeroed slack in front of (FS) and behind the padded planes of a batch.
MFL_HD int diag_front_slack(int k) { return ((2 * k + 2 + 15) / 16) * 16; }

// Channel split: CH channels per wave, NCS waves per group, NGW groups per workgroup.  The
// accumulator tile of a thread is CH x k registers, so wide windows get narrow channel slices.
// Returns false when no built configuration fits.
inline bool diag_split(int C, int k, int &CH, int &NCS, int &NGW)
{
    if (k >= 9) {
        CH = 5;
        NCS = (C + 4) / 5;
        if (NCS > 4) return false;
        NGW = NCS == 1 ? 8 : NCS == 2 ? 4 : 2;
        return true;
    }
    if (C <= 8) { NCS = 1; NGW = 8; CH = 8; return true; }
    NCS = C <= 22 ? 2 : 3;
    NGW = NCS == 2 ? 4 : 2;
#ifdef MFL_EXP_NGW
    if (NCS == 2) NGW = MFL_EXP_NGW;   // timing experiment: smaller workgroups, more of them per CU
#endif
    const int need = (C + NCS - 1) / NCS;
    CH = need <= 8 ? 8 : need <= 10 ? 10 : 11;
    return true;
}

// cb: channels one launch computes (all C of them unless the channels are computed block by block)
inline bool diag_supported(int C, int H, int W, int k, int cb = 0)
{
    if (cb <= 0) cb = C;
    if (!(k == 3 || k == 5 || k == 7 || k == 9 || k == 11)) return false;
    if (cb > 33 || cb > C || W < 64 + k || W >= 16384) return false;
    const long long CK = (long long)C * k * k, L = (long long)H * W;
    if (L < 4 * CK) return false;   // too few source pixels per row of the scrambled matrix to fill a wave
    int ch, ncs, ngw;
    return diag_split(cb, k, ch, ncs, ngw);
}

// Width of the channel blocks for C > 33 labels: the fewest blocks of at most 22 channels (two waves of 11 per
// group and four groups per workgroup, the fastest split measured), equal widths.
inline int diag_block_width(int C)
{
    const int nblk = (C + 21) / 22;
    return (C + nblk - 1) / nblk;
}

// Table record of one (b-block, group, tap, channel split), in dwords (at most 64, so that one
// vector load puts a whole record into one register, lane j = dword j):
//   [0, CH)              byte offsets of A        (channels past C repeat the first channel of the split)
//   [CH, 2CH)            byte offsets of B
//   [2CH]                header: bit i (i < CH) step i is flagged; bits 16..18 number of slots;
//                        bit 31 more flagged steps than slots (the per-step columns below are needed)
//   [2CH+1, +slots)      slots, one dword per flagged step (at most kDiagSlots):
//                          step | (TA + 64) << 8 | (TB + 64) << 16
//   [xw base, ...)       16-bit pairs: xA | (xB != xA) << 15   (lane-0 source columns, for the steps
//                        past the slots)
//   [kyw base, ...)      one byte per step: ky' of A | ky' of B << 4, the window row of the line (c', ky') the
//                        side's element belongs to.  An element of pad(x) is shared by k lines (one per window
//                        row that reaches it), and in each line by the side A of one lane and the side B of
//                        another: (side, ky') names which of its 2k contributors a lane is, which lets the
//                        backward store what it sends to an element instead of adding it atomically.
// A flagged side is described by a threshold T (diag_side): with p = lane - T, pixel n of the lane
// reads the table's element ("shared") if p < n and the element gap floats further ("extra") if
// p >= n.  Where the run wraps to the next image row at lane T (1 <= T <= 63) the table holds the
// plain offset; where the row started -T <= k-2 pixels before lane 0 it holds the offset moved back
// by the gap, so that one rule serves both; a side that meets no row start has T = 64 (p < 0 <= n).
constexpr int kDiagSlots = 3;
MFL_HD int diag_slot_base(int CH) { return 2 * CH + 1; }
MFL_HD int diag_xw_base(int CH) { return 2 * CH + 1 + kDiagSlots; }
MFL_HD int diag_kyw_base(int CH) { return diag_xw_base(CH) + (CH + 1) / 2; }
MFL_HD int diag_recw(int CH) { return ((diag_kyw_base(CH) + (CH + 3) / 4 + 3) / 4) * 4; }

// A last b-block with at most this many live lanes is not run by the diagonal kernels (unless the
// caller asks for all blocks: the backward does, its tail pixels would scatter one 4-byte atomic per
// lane to unrelated lines, which costs far more than the nearly empty workgroups).
constexpr int kDiagTailLanes = 16;

// Returns 0 on success, -1 invalid, -3 too large for 32-bit indexing.
inline int make_diag_geom(DiagGeom &g, int B, int C, int H, int W, int k, int lds_budget_bytes, bool allow_tail = true, int cb = 0)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return -1;
    if (cb <= 0) cb = C;
    if (!diag_supported(C, H, W, k, cb)) return -1;
    const long long K = (long long)k * k, L = (long long)H * W, CK = (long long)C * K;
    const long long Hp = (long long)H + k - 1, Wp = (long long)W + k - 1, plane = Hp * Wp;
    if (L >= (1LL << 30) || (C * plane + 4 * Wp + 256) * 4 >= (1LL << 31)) return -3;
    // the 16-byte pixels of the padded image are reached through 32-bit byte offsets as well (few channels,
    // very large images: the planes pass the check above and the image does not)
    if ((plane + (long long)k * Wp + 64) * 16 >= (1LL << 31)) return -3;
    g.B = B; g.C = C; g.H = H; g.W = W; g.k = k; g.K = (int)K; g.p0 = k / 2;
    g.c0 = 0; g.Cb = cb; g.Cv = cb;
    g.L = (int)L; g.CK = (int)CK; g.Hp = (int)Hp; g.Wp = (int)Wp; g.plane = (int)plane; g.gap = k - 1;
    g.NB = (int)((L + CK - 1) / CK);
    g.NG = C * k;
    g.BL = 64;
    g.NBB = (g.NB + k - 1 + 63) / 64;
    g.tail_lanes = 0;
    {
        // a last block with only a few live lanes costs a whole workgroup for a handful of pixels:
        // leave it to the per-pixel tail kernel
        const int rem = g.NB + k - 1 - 64 * (g.NBB - 1);
        if (allow_tail && g.NBB > 1 && rem <= kDiagTailLanes) {
            g.tail_lanes = rem;
            g.NBB -= 1;
        }
    }
    diag_split(cb, k, g.CH, g.NCS, g.NGW);
    g.NGB = (g.NG + g.NGW - 1) / g.NGW;
    g.RUN = g.NGW * k;
    g.RUNP = (g.RUN & 1) ? g.RUN + 2 : g.RUN + 1;
    g.NBR = 64 + k - 1;
    g.Lk = (int)(L % k);
    g.FS = diag_front_slack(k);
    g.RECW = diag_recw(g.CH);
    int tb = k;
    while (tb > 1 && 2LL * tb * g.NBR * g.RUNP * 4 > lds_budget_bytes) --tb;
    g.TB = tb;
    const int chunks = (k + tb - 1) / tb;
    g.ngroups = k * chunks;
    const long long recs = (long long)g.NBB * g.NG * K * g.NCS;
    if (recs * g.RECW >= (1LL << 31)) return -3;
    return 0;
}

MFL_HD size_t diag_table_dwords(const DiagGeom &g)
{
    return (size_t)g.NBB * g.NG * g.K * g.NCS * g.RECW;
}

MFL_HD size_t diag_record_index(const DiagGeom &g, int beta, int gamma, int t, int ci)
{
    return ((((size_t)beta * g.NG + gamma) * g.K + t) * g.NCS + ci) * g.RECW;
}

// One (b-block, group, tap, channel) step of the gather:
//   offA / offB  float index, relative to FS floats before the image's padded planes, of the
//                element lane 0 shares among its pixels with s + n < k (A) / s + n >= k (B);
//   xA / xB      image column of lane 0's source pixel for A / B;
//   flagged      the 64-lane run meets an image-row start: the per-lane corrected path is needed.
struct DiagStep {
    int offA, offB, xA, xB;
    int kyA, kyB;             // window row of the line A / B belongs to
    bool flagged;
};

MFL_HD DiagStep diag_step(const DiagGeom &g, int beta, int gamma, int t, int c)
{
    const long long G = ((long long)c * g.K + t) * g.L + (long long)gamma * g.k;
    const long long lp0 = G / g.CK;
    const int r0 = (int)(G - lp0 * g.CK);
    const int rho = r0 / g.k, s = r0 - rho * g.k;
    DiagStep st;
    // A: line rho of source pixel lp0 + 64*beta
    const long long uA = lp0 + (long long)g.BL * beta;
    const int yA = (int)(uA / g.W);
    st.xA = (int)(uA - (long long)yA * g.W);
    const int cA = rho / g.k, kyA = rho - cA * g.k;
    st.offA = g.FS + cA * g.plane + (yA + kyA) * g.Wp + st.xA + s;
    st.kyA = kyA;
    // B: the next line, possibly of the next source pixel
    long long uB = uA;
    int rhoB = rho + 1;
    if (rhoB == g.C * g.k) { rhoB = 0; uB = uA + 1; }
    const int yB = (int)(uB / g.W);
    st.xB = (int)(uB - (long long)yB * g.W);
    const int cB = rhoB / g.k, kyB = rhoB - cB * g.k;
    st.offB = g.FS + cB * g.plane + (yB + kyB) * g.Wp + st.xB + s - g.k;
    st.kyB = kyB;
    const bool okA = st.xA >= g.k - 1 && st.xA + 63 <= g.W - 1;
    const bool okB = st.xB >= g.k - 1 && st.xB + 63 <= g.W - 1;
    st.flagged = !(okA && okB);
    return st;
}

// Float index (same base as offA / offB) that lane `lane`, pixel n reads in a step whose side
// (A or B) has offset `off` and lane-0 column `x0`.  The unflagged case reduces to off + lane.
MFL_HD int diag_lane_index(const DiagGeom &g, int off, int x0, int lane, int n)
{
    int pos = x0 + lane;
    const bool wrapped = pos >= g.W;
    if (wrapped) pos -= g.W;
    return off + lane + (wrapped ? g.gap : 0) - (pos < n ? g.gap : 0);
}

// Alignment of tap column kx: pixels n < k - s read A, the others B.
MFL_HD int diag_align(const DiagGeom &g, int kx) { return (int)(((long long)kx * g.Lk) % g.k); }

// Weight groups of one iteration: tap columns ordered by alignment class, each split into chunks of
// at most TB taps.  Packed as s | kx << 8 | ky0 << 16 | nky << 24.
MFL_HD int diag_groups(const DiagGeom &g, uint32_t *out)
{
    int n = 0;
    for (int s = 0; s < g.k; ++s)
        for (int kx = 0; kx < g.k; ++kx) {
            if (diag_align(g, kx) != s) continue;
            for (int ky0 = 0; ky0 < g.k; ky0 += g.TB) {
                const int nky = (g.k - ky0 < g.TB) ? g.k - ky0 : g.TB;
                out[n++] = (uint32_t)s | ((uint32_t)kx << 8) | ((uint32_t)ky0 << 16) | ((uint32_t)nky << 24);
            }
        }
    return n;
}

// One side of a step whose lane-0 element has float index `off` and source column x0: the threshold
// T and the float index the table holds (see the record layout above).
MFL_HD void diag_side(const DiagGeom &g, int off, int x0, int &T, int &off_table)
{
    if (x0 + 63 > g.W - 1) { T = g.W - x0; off_table = off; }
    else if (x0 < g.k - 1) { T = -x0; off_table = off - g.gap; }
    else { T = 64; off_table = off; }
}

// Fill one table record (all CH steps of one (beta, gamma, t, channel split)); layout: diag_recw.
MFL_HD void diag_fill_record(const DiagGeom &g, int beta, int gamma, int t, int ci, uint32_t *rec)
{
    uint32_t flags = 0;
    int nslots = 0, nflagged = 0;
    const int sb = diag_slot_base(g.CH), xb = diag_xw_base(g.CH), kb = diag_kyw_base(g.CH);
    for (int i = 0; i < g.RECW; ++i) rec[i] = 0;
    for (int i = 0; i < g.CH; ++i) {
        int c = ci * g.CH + i;
        const bool dummy = c >= g.Cv;
        if (dummy) c = ci * g.CH < g.Cv ? ci * g.CH : 0;
        const DiagStep st = diag_step(g, beta, gamma, t, g.c0 + c);
        int tA, tB, oA, oB;
        diag_side(g, st.offA, st.xA, tA, oA);
        diag_side(g, st.offB, st.xB, tB, oB);
        rec[i] = (uint32_t)oA * 4u;
        rec[g.CH + i] = (uint32_t)oB * 4u;
        const uint32_t x16 = (uint32_t)st.xA | ((st.xB != st.xA) ? 0x8000u : 0u);
        rec[xb + i / 2] |= x16 << (16 * (i & 1));
        rec[kb + i / 4] |= ((uint32_t)st.kyA | ((uint32_t)st.kyB << 4)) << (8 * (i & 3));
        if (st.flagged && !dummy) {
            flags |= 1u << i;
            ++nflagged;
            if (nslots < kDiagSlots)
                rec[sb + nslots++] = (uint32_t)i | ((uint32_t)(tA + 64) << 8) | ((uint32_t)(tB + 64) << 16);
        }
    }
    rec[2 * g.CH] = flags | ((uint32_t)nslots << 16) | (nflagged > kDiagSlots ? 0x80000000u : 0u);
}

// Whether pixel n of lane `lane` reads the extra element of a side with threshold T.
MFL_HD bool diag_reads_extra(int T, int lane, int n) { return lane - T >= n; }

// Pixels of the b-block left to the tail kernel.  Pixel n of lane `lane` of that block has
// b = 64 * NBB + lane - n, a pixel only while b < NB: diag_tail_lanes(n) lanes per n.  Candidate idx
// in [0, NG * diag_tail_per_group) is (group, n, lane); diag_tail_pixel returns the output pixel l,
// or -1 for a candidate past the end of the image (l >= L).
MFL_HD int diag_tail_lanes(const DiagGeom &g, int n)
{
    const int m = g.NB - g.BL * g.NBB + n;
    return m < 0 ? 0 : (m < g.tail_lanes ? m : g.tail_lanes);
}
MFL_HD int diag_tail_per_group(const DiagGeom &g)
{
    int s = 0;
    for (int n = 0; n < g.k; ++n) s += diag_tail_lanes(g, n);
    return s;
}
MFL_HD long long diag_tail_candidates(const DiagGeom &g) { return (long long)g.NG * diag_tail_per_group(g); }
MFL_HD int diag_tail_pixel(const DiagGeom &g, long long idx)
{
    const int per = diag_tail_per_group(g);
    const int gamma = (int)(idx / per);
    int r = (int)(idx - (long long)gamma * per), n = 0;
    while (r >= diag_tail_lanes(g, n)) r -= diag_tail_lanes(g, n++);
    const long long a = (long long)gamma * g.k + n, bb = (long long)g.BL * g.NBB + r - n;
    const long long l = a + bb * g.CK;
    return l < g.L ? (int)l : -1;
}

}  // namespace mfl

#endif  // MFL_DIAG_H
