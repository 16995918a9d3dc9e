// Geometry, tap stream and address table of the "ring" gather kernels (mfl_ring.hip): the diagonal
// ownership of mfl_diag.h (a thread owns k output pixels that share one element of pad(x) per
// (channel, tap)), with the shared elements staged through LDS instead of registers.  Plain C++
// shared by the device code and by the host emulation in the CPU test-suite.
//
// Arithmetic restated (not code) from Conv2dLocal.forward, LocalConvolution.py:59-90; the index map
// along its diagonals, sides A / B, the alignment class of a tap column and the rule for image-row
// starts (threshold T, "shared" and "extra" element): header comment and diag_side in mfl_diag.h.
//
// ---- what changes against the first diagonal kernels ---------------------------------------------
// Every wave is a stream of its own: it takes the tap weights of its own 64 x k pixels from the call's weight
// table, reads one table record per tap with scalar loads, and keeps the
// shared elements (A, B) of the next tap in flight in registers that are refilled in place.  A step
// that meets an image-row start needs, per side, the element gap floats further ("extra") for a few
// lanes: that is the element lane + gap of the same load, fetched with a lane permute.  For it to
// exist for every lane that owns pixels, only the first BL = 65 - k lanes of a wave own pixels
// (b = BL*beta + lane - n); the last k - 1 lanes load along and their accumulators are dropped.
//
// ---- tap stream -------------------------------------------------------------------------------
// The taps of an iteration are walked in one fixed order (the stream): tap columns sorted by
// alignment class (which pixels of a lane take side A is then fixed per run of taps, and each class
// has its own unrolled loop body), rows top to bottom.  Stream position i holds tap ring_stream_tap(i).
//
// ---- table record of one (b-block, group, stream position i = 0 .. K, channel split) -------------
//   [0, CH)        scalar byte offset of side A of the tap at position i (position K repeats K-1)
//   [CH, 2CH)      the same for side B
//   [2CH, ...)     16 bits per step: (TA + 64) | (TB + 64) << 8; a step with both thresholds 64 meets
//                  no image-row start
//   then one word  bit s: step s of the NEXT stream position meets an image-row start, so the lanes past BL
//                  must load along for it (their elements are the "extra" ones); otherwise only the BL owning
//                  lanes request elements (a shorter run touches fewer cache lines)
#ifndef MFL_RING_H
#define MFL_RING_H

#include "mfl_diag.h"

namespace mfl {


struct RingGeom {
    DiagGeom d;                               // B .. FS as in the diagonal kernels; CH/NCS/NGW/RECW of the ring split
    int recs_per_wave;                        // K + 1
    unsigned long long cols_kx;               // tap columns in stream order, 4 bits each
    unsigned long long cols_sa;               // their alignment classes (non-decreasing), 4 bits each; 15: none
};

// Channel split of the ring kernels: CH channels per wave (CH * k accumulators per lane), NCS waves
// per group, NGW groups per workgroup (four to six waves per workgroup, so that two workgroups share a CU and
// the epilogue of one overlaps the tap loop of the other).  Wider channel slices amortise a tap's weights,
// record and loop overhead over more multiply-adds: 10 channels per wave where that saves a wave per group
// (19 labels: 2 x 10 instead of 3 x 7; measured 8 % faster per step at 1024 x 2048, 11 x 11).
inline bool ring_split(int C, int k, int &CH, int &NCS, int &NGW)
{
    const int narrow = k == 9 ? 8 : 7;       // CH * k accumulators per lane: 72 and 77
    const int ncs_narrow = (C + narrow - 1) / narrow, ncs_wide = (C + 9) / 10;
    CH = ncs_wide < ncs_narrow ? 10 : narrow;
    NCS = ncs_wide < ncs_narrow ? ncs_wide : ncs_narrow;
    if (NCS < 2) NCS = 2;                    // (a workgroup has at least four waves)
    if (NCS > 4) return false;
    NGW = NCS == 4 ? 1 : 2;
    return true;
}

// The stream kernels are built for the wide windows (9 x 9, 11 x 11); measured on 21 labels, 512 x 512,
// 7 x 7, the diagonal kernels of mfl_diag.hip are as fast per step and need no weight table.
inline bool ring_supported(int C, int H, int W, int k)
{
    if (!(k == 9 || k == 11)) return false;
    if (C > 32) return false;                // the fused epilogue holds the label matrix as one 32 x 32 tile
    if (W < 64 + k || W >= 16384) return false;
    const long long CK = (long long)C * k * k, L = (long long)H * W;
    if (L < 4 * CK) return false;
    int ch, ncs, ngw;
    return ring_split(C, k, ch, ncs, ngw);
}

MFL_HD int ring_recw(int CH) { return ((2 * CH + (CH + 1) / 2 + 1 + 3) / 4) * 4; }

// Returns 0 on success, -1 invalid / unsupported, -3 too large for 32-bit indexing.
inline int make_ring_geom(RingGeom &r, int B, int C, int H, int W, int k)
{
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return -1;
    if (!ring_supported(C, H, W, k)) return -1;
    DiagGeom &g = r.d;
    const long long K = (long long)k * k, L = (long long)H * W, CK = (long long)C * K;
    const long long Hp = (long long)H + k - 1, Wp = (long long)W + k - 1, plane = Hp * Wp;
    // fp32 planes of one image and the 16-byte pixels of the padded image, in bytes, through 32-bit offsets
    if (L >= (1LL << 30) || (C * plane + 4 * Wp + 256) * 4 >= (1LL << 31)) return -3;
    if ((plane + (long long)k * Wp + 64) * 16 >= (1LL << 31)) return -3;
    g.B = B; g.C = C; g.H = H; g.W = W; g.k = k; g.K = (int)K; g.p0 = k / 2;
    g.c0 = 0; g.Cb = C; g.Cv = C;
    g.L = (int)L; g.CK = (int)CK; g.Hp = (int)Hp; g.Wp = (int)Wp; g.plane = (int)plane; g.gap = k - 1;
    g.NB = (int)((L + CK - 1) / CK);
    g.NG = C * k;
    // the last k - 1 lanes of a wave own no pixels: what they load is the "extra" element (gap floats
    // further) of the lanes before them, fetched with a lane permute at image-row starts
    g.BL = 65 - k;
    g.NBB = (g.NB + k - 1 + g.BL - 1) / g.BL;
    g.tail_lanes = 0;
    {
        const int rem = g.NB + k - 1 - g.BL * (g.NBB - 1);
        if (g.NBB > 1 && rem <= kDiagTailLanes) {
            g.tail_lanes = rem;
            g.NBB -= 1;
        }
    }
    ring_split(C, k, g.CH, g.NCS, g.NGW);
    g.NGB = (g.NG + g.NGW - 1) / g.NGW;
    g.RUN = g.NGW * k;
    g.RUNP = (g.RUN & 1) ? g.RUN + 2 : g.RUN + 1;
    g.NBR = g.BL + k - 1;
    g.Lk = (int)(L % k);
    g.FS = diag_front_slack(k);
    g.RECW = ring_recw(g.CH);
    g.TB = 1;
    g.ngroups = (int)K;
    r.recs_per_wave = (int)K + 1;
    int n = 0;
    r.cols_kx = 0;
    r.cols_sa = ~0ULL;
    for (int s = 0; s < k; ++s)
        for (int kx = 0; kx < k; ++kx)
            if (diag_align(g, kx) == s) {
                r.cols_kx |= (unsigned long long)kx << (4 * n);
                r.cols_sa = (r.cols_sa & ~(15ULL << (4 * n))) | ((unsigned long long)s << (4 * n));
                ++n;
            }
    const long long recs = (long long)g.NBB * g.NG * (K + 1) * g.NCS;
    if (recs * g.RECW >= (1LL << 31)) return -3;
    return 0;
}

// Floats of the weight table of a call: per (image, b-block, group) and stream position, 64 lanes x
// ceil(k / 4) pieces of 4 floats.
MFL_HD size_t ring_weight_floats(const RingGeom &r)
{
    return (size_t)r.d.B * r.d.NBB * r.d.NG * r.d.K * (size_t)(((r.d.k + 3) / 4) * 256);
}

MFL_HD size_t ring_table_dwords(const RingGeom &r)
{
    return (size_t)r.d.NBB * r.d.NG * r.recs_per_wave * r.d.NCS * r.d.RECW;
}

// Record of stream position i (0 .. K).
MFL_HD size_t ring_record_index(const RingGeom &r, int beta, int gamma, int i, int ci)
{
    return ((((size_t)beta * r.d.NG + gamma) * r.recs_per_wave + i) * r.d.NCS + ci) * r.d.RECW;
}

// Tap (ky * k + kx) at stream position i.
MFL_HD int ring_stream_tap(const RingGeom &r, int i)
{
    const int col = i / r.d.k, ky = i - col * r.d.k;
    return ky * r.d.k + (int)((r.cols_kx >> (4 * col)) & 15);
}

MFL_HD void ring_fill_record(const RingGeom &r, int beta, int gamma, int i, int ci, uint32_t *rec)
{
    const DiagGeom &g = r.d;
    for (int j = 0; j < g.RECW; ++j) rec[j] = 0;
    const int t = ring_stream_tap(r, i < g.K ? i : g.K - 1);
    for (int s = 0; s < g.CH; ++s) {
        int c = ci * g.CH + s;
        if (c >= g.C) c = ci * g.CH < g.C ? ci * g.CH : 0;      // dummy channels repeat the split's first one
        const DiagStep st = diag_step(g, beta, gamma, t, c);
        int tA, tB, oA, oB;
        diag_side(g, st.offA, st.xA, tA, oA);
        diag_side(g, st.offB, st.xB, tB, oB);
        rec[s] = (uint32_t)oA * 4u;
        rec[g.CH + s] = (uint32_t)oB * 4u;
        const uint32_t t16 = (uint32_t)(tA + 64) | ((uint32_t)(tB + 64) << 8);
        rec[2 * g.CH + s / 2] |= t16 << (16 * (s & 1));
    }
    // which steps of the next position (position K repeats K - 1) meet an image-row start
    uint32_t next = ~0u;
    if (i < g.K) {
        next = 0;
        const int tn = ring_stream_tap(r, i + 1 < g.K ? i + 1 : g.K - 1);
        for (int s = 0; s < g.CH; ++s) {
            int c = ci * g.CH + s;
            if (c >= g.C) c = ci * g.CH < g.C ? ci * g.CH : 0;
            const DiagStep st = diag_step(g, beta, gamma, tn, c);
            int tA, tB, oA, oB;
            diag_side(g, st.offA, st.xA, tA, oA);
            diag_side(g, st.offB, st.xB, tB, oB);
            if (tA != 64 || tB != 64) next |= 1u << s;
        }
    }
    rec[2 * g.CH + (g.CH + 1) / 2] = next;
}

}  // namespace mfl

#endif  // MFL_RING_H
