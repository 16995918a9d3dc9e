// Host emulation of the diagonal gather (semantic_segmentation_amd/csrc/mfl_diag.h): the address
// table is built with the same functions the device uses and every (workgroup, wave, lane, pixel)
// of the device kernel is replayed as loop iterations on the CPU, reading the padded planes only
// through table offsets and the per-lane index rule.  It checks ownership (every pixel computed
// exactly once), the A/B split per tap column, flagged steps and tails without a GPU.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../semantic_segmentation_amd/csrc/mfl_diag.h"

using namespace mfl;

// y = x * window sums (Conv2dLocal.forward with the colour distance of img).  count[b][l] receives
// how many times each output pixel was produced (must be exactly 1).  Returns 0 on success.
extern "C" int emu_diag_local_forward(const float *x, const float *img, float pw, float cw, float *y, int *count, int B,
                                      int C, int H, int W, int k)
{
    DiagGeom g;
    int rc = make_diag_geom(g, B, C, H, W, k, 96 * 1024);
    if (rc) return rc;
    const size_t per_image = (size_t)C * g.plane;
    const size_t npad = (size_t)g.FS + B * per_image + 3 * g.Wp + 64;
    std::vector<float> buf(npad, 0.f);
    float *xp = buf.data() + g.FS;
    for (int bc = 0; bc < B * C; ++bc)
        for (int yy = 0; yy < H; ++yy)
            memcpy(&xp[((size_t)bc * g.Hp + yy + g.p0) * g.Wp + g.p0], x + ((size_t)bc * H + yy) * W, sizeof(float) * W);
    std::vector<uint32_t> table(diag_table_dwords(g));
    for (int beta = 0; beta < g.NBB; ++beta)
        for (int gamma = 0; gamma < g.NG; ++gamma)
            for (int t = 0; t < g.K; ++t)
                for (int ci = 0; ci < g.NCS; ++ci)
                    diag_fill_record(g, beta, gamma, t, ci, &table[diag_record_index(g, beta, gamma, t, ci)]);
    std::vector<uint32_t> groups(g.k * g.k);
    const int ngroups = diag_groups(g, groups.data());
    if (ngroups != g.ngroups) return -10;

    memset(y, 0, sizeof(float) * (size_t)B * C * g.L);
    memset(count, 0, sizeof(int) * (size_t)B * g.L);
    const size_t Lz = (size_t)g.L;
    for (int b = 0; b < B; ++b) {
        const float *base = buf.data() + (size_t)b * per_image;   // FS floats before image b's planes
        const float *im = img + (size_t)b * 3 * Lz;
        for (int beta = 0; beta < g.NBB; ++beta)
            for (int gamma = 0; gamma < g.NG; ++gamma)
                for (int lane = 0; lane < 64; ++lane)
                    for (int n = 0; n < g.k; ++n) {
                        const long long a = (long long)gamma * g.k + n, bb = 64LL * beta + lane - n;
                        const long long l = a + bb * g.CK;
                        if (bb < 0 || l >= g.L) continue;
                        const int py = (int)(l / W), px = (int)(l % W);
                        count[(size_t)b * Lz + l] += 1;
                        for (int ci = 0; ci < g.NCS; ++ci)
                            for (int i = 0; i < g.CH; ++i) {
                                const int c = ci * g.CH + i;
                                if (c >= C) continue;
                                float acc = 0.f;
                                for (int gi = 0; gi < ngroups; ++gi) {
                                    const int s = groups[gi] & 255, kx = (groups[gi] >> 8) & 255;
                                    const int ky0 = (groups[gi] >> 16) & 255, nky = groups[gi] >> 24;
                                    for (int ky = ky0; ky < ky0 + nky; ++ky) {
                                        const int t = ky * k + kx;
                                        const uint32_t *rec = &table[diag_record_index(g, beta, gamma, t, ci)];
                                        const uint32_t hdr = rec[2 * g.CH];
                                        const bool flagged = (hdr >> i) & 1;
                                        const bool useA = s + n < k;
                                        const int off = (int)(rec[useA ? i : g.CH + i] / 4);
                                        int idx = off + lane;
                                        if (flagged) {
                                            // threshold of the side: from the step's slot, or (steps past
                                            // the slots of an overflow record) from its column word as the
                                            // device derives it
                                            const int nslots = (hdr >> 16) & 7;
                                            int T = 1 << 20;
                                            for (int j = 0; j < nslots; ++j) {
                                                const uint32_t d = rec[diag_slot_base(g.CH) + j];
                                                if ((int)(d & 15) == i) T = (int)((d >> (useA ? 8 : 16)) & 255) - 64;
                                            }
                                            const uint32_t x16 =
                                                (rec[diag_xw_base(g.CH) + i / 2] >> (16 * (i & 1))) & 0xffff;
                                            const int xA = x16 & 0x7fff;
                                            int xB = xA;
                                            if (x16 & 0x8000) xB = (xA + 1 == W) ? 0 : xA + 1;
                                            const int x0 = useA ? xA : xB;
                                            const int Tdev = x0 + 63 > W - 1 ? W - x0 : (x0 < k - 1 ? -x0 : 64);
                                            if (T == (1 << 20)) {
                                                if (!(hdr & 0x80000000u)) return -11;
                                                T = Tdev;
                                            } else if (T != Tdev) {
                                                return -12;
                                            }
                                            if (diag_reads_extra(T, lane, n)) idx += g.gap;
                                            // against the direct statement of the rule
                                            const int off0 = (T <= 0) ? off + g.gap : off;
                                            if (idx != diag_lane_index(g, off0, x0, lane, n)) return -13;
                                        }
                                        // tap weight
                                        const int ny = py + ky - g.p0, nx = px + kx - g.p0;
                                        const bool in = ny >= 0 && ny < H && nx >= 0 && nx < W;
                                        float cd = 0.f;
                                        for (int ch = 0; ch < 3; ++ch) {
                                            const float d = im[ch * Lz + l] - (in ? im[ch * Lz + (size_t)ny * W + nx] : 0.f);
                                            cd += d * d;
                                        }
                                        const int dy = ky - g.p0, dx = kx - g.p0;
                                        const float wgt = (float)(dy * dy + dx * dx) * pw + cd * cw;
                                        acc += wgt * base[idx];
                                    }
                                }
                                y[((size_t)b * C + c) * Lz + l] = acc * x[((size_t)b * C + c) * Lz + l];
                            }
                    }
        // the pixels of a nearly empty last b-block are enumerated for the tail kernel: computed here
        // directly from the definition (the device tail kernel does the same per pixel)
        for (long long idx = 0; g.tail_lanes && idx < diag_tail_candidates(g); ++idx) {
            const int l = diag_tail_pixel(g, idx);
            if (l < 0) continue;
            count[(size_t)b * Lz + l] += 1;
            const int py = l / W, px = l % W;
            const float *xpad = buf.data() + g.FS + (size_t)b * per_image;
            for (int c = 0; c < C; ++c) {
                float acc = 0.f;
                for (int t = 0; t < g.K; ++t) {
                    const int ky = t / k, kx = t % k;
                    const long long gg = ((long long)c * g.K + t) * g.L + l;
                    const long long lp = gg / g.CK;
                    const int rr = (int)(gg - lp * g.CK);
                    const int cs = rr / g.K, ts = rr % g.K;
                    const int ys = (int)(lp / W), xs = (int)(lp % W);
                    const float val = xpad[(size_t)cs * g.plane + (ys + ts / k) * g.Wp + xs + ts % k];
                    const int ny = py + ky - g.p0, nx = px + kx - g.p0;
                    const bool in = ny >= 0 && ny < H && nx >= 0 && nx < W;
                    float cd = 0.f;
                    for (int ch = 0; ch < 3; ++ch) {
                        const float d = im[ch * Lz + l] - (in ? im[ch * Lz + (size_t)ny * W + nx] : 0.f);
                        cd += d * d;
                    }
                    const int dy = ky - g.p0, dx = kx - g.p0;
                    acc += ((float)(dy * dy + dx * dx) * pw + cd * cw) * val;
                }
                y[((size_t)b * C + c) * Lz + l] = acc * x[((size_t)b * C + c) * Lz + l];
            }
        }
    }
    return 0;
}

// Status of the geometry alone (0, -1 unsupported, -3 too large for 32-bit offsets): no table, no data.
extern "C" int emu_diag_geom_status(int B, int C, int H, int W, int k)
{
    DiagGeom g;
    return make_diag_geom(g, B, C, H, W, k, 96 * 1024);
}

// The backward scatter's store targets (diag_kernel MODE 3, mfl_diag.hip), replayed from the table records: per
// (b-block, group, tap, channel, lane) the copy index (side, window row of the record's ky byte) and the float index
// of the shared and of the extra element that receive a sum.  Counts the targets with more than one writer; a lane is
// an edge lane by the kernel's rule (within (k-1)(CK-1) pixels of either end of the pixel range).
// out[0] = targets with two or more writers, out[1] = of those, targets with a writer that is no edge lane,
// out[2] = largest number of writers of one target, out[3] = targets written.
extern "C" int emu_diag_scatter_targets(int C, int H, int W, int k, long long *out)
{
    DiagGeom g;
    int rc = make_diag_geom(g, 1, C, H, W, k, 96 * 1024, false);
    if (rc) return rc;
    struct Writers { int n; bool non_edge; };
    std::vector<std::vector<Writers>> seen(2 * k, std::vector<Writers>((size_t)g.FS + (size_t)C * g.plane + 3 * g.Wp + 64, Writers{0, false}));
    std::vector<uint32_t> rec(g.RECW);
    const long long span = (long long)(k - 1) * (g.CK - 1);
    auto threshold = [&](int x0) { return x0 + 63 > W - 1 ? W - x0 : (x0 < k - 1 ? -x0 : 64); };
    for (int beta = 0; beta < g.NBB; ++beta)
        for (int gamma = 0; gamma < g.NG; ++gamma)
            for (int t = 0; t < g.K; ++t) {
                const int NA = k - diag_align(g, t % k);
                for (int ci = 0; ci < g.NCS; ++ci) {
                    diag_fill_record(g, beta, gamma, t, ci, rec.data());
                    const uint32_t hdr = rec[2 * g.CH];
                    for (int i = 0; i < g.CH; ++i) {
                        if (ci * g.CH + i >= C) continue;
                        const uint32_t oA = rec[i] >> 2, oB = rec[g.CH + i] >> 2;
                        const uint32_t kyw = rec[diag_kyw_base(g.CH) + (i >> 2)] >> (8 * (i & 3));
                        const int copyA = kyw & 15, copyB = k + ((kyw >> 4) & 15);
                        const bool flagged = (hdr >> i) & 1;
                        const uint32_t x16 = (rec[diag_xw_base(g.CH) + (i >> 1)] >> (16 * (i & 1))) & 0xffffu;
                        const int xA = x16 & 0x7fff;
                        const int xB = (x16 & 0x8000) ? (xA + 1 == W ? 0 : xA + 1) : xA;
                        for (int lane = 0; lane < 64; ++lane) {
                            const long long l_first = (long long)gamma * k + ((long long)64 * beta + lane) * g.CK;
                            const long long l_last = l_first + (k - 1) - (long long)(k - 1) * g.CK;
                            const bool edge = l_last < span || l_first >= g.L - span;
                            const int pA = lane - threshold(xA), pB = lane - threshold(xB);
                            bool w[4] = {false, false, false, false};      // shared A, extra A, shared B, extra B
                            for (int n = 0; n < k; ++n) {
                                const long long l = l_first + n - (long long)n * g.CK;
                                if (l < 0 || l >= g.L) continue;          // D = 0: nothing is sent
                                if (n < NA) w[(flagged && pA >= n) ? 1 : 0] = true;
                                else w[(flagged && pB >= n) ? 3 : 2] = true;
                            }
                            auto put = [&](int copy, size_t idx) {
                                Writers &s = seen[copy][idx];
                                s.n += 1;
                                s.non_edge = s.non_edge || !edge;
                            };
                            if (w[0]) put(copyA, (size_t)oA + lane);
                            if (w[1]) put(copyA, (size_t)oA + lane + g.gap);
                            if (w[2]) put(copyB, (size_t)oB + lane);
                            if (w[3]) put(copyB, (size_t)oB + lane + g.gap);
                        }
                    }
                }
            }
    out[0] = out[1] = out[2] = out[3] = 0;
    for (auto &copy : seen)
        for (auto &s : copy) {
            if (s.n == 0) continue;
            out[3] += 1;
            if (s.n > out[2]) out[2] = s.n;
            if (s.n > 1) { out[0] += 1; if (s.non_edge) out[1] += 1; }
        }
    return 0;
}
