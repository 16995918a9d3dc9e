// Host emulation of the stream gather kernels (semantic_segmentation_amd/csrc/mfl_ring.h, mfl_ring.hip): the
// records are built with the same functions the device uses and every (b-block, group, lane, pixel, stream
// position, channel) of the device loop is replayed on the CPU.  A lane reads the padded planes only at
// "record offset + lane" (the shared element of a side) and, where the record's threshold says so, takes
// instead the shared element of lane + gap of the same 64-lane load (the device's lane permute).  Checks
// without a GPU: every pixel is produced exactly once (regular tiles plus the tail pixels of a nearly empty
// last b-block), the side split per alignment class, the thresholds, that the 65 - k owning lanes never
// need an element past the 64 loaded ones, and that an element of a lane past BL is only ever needed in a step
// the previous record's flag word announces (those lanes request nothing otherwise).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../semantic_segmentation_amd/csrc/mfl_ring.h"

using namespace mfl;

extern "C" int emu_ring_supported(int C, int H, int W, int k) { return ring_supported(C, H, W, k) ? 1 : 0; }

// y = x * window sums (Conv2dLocal.forward with the colour distance of img); count[b][l] = how many times the
// pixel was produced.  Returns 0, a make_ring_geom status, or -20.. for a violated invariant.
extern "C" int emu_ring_local_forward(const float *x, const float *img, float pw, float cw, float *y, int *count, int B, int C,
                                      int H, int W, int k)
{
    RingGeom r;
    const int rc = make_ring_geom(r, B, C, H, W, k);
    if (rc) return rc;
    const DiagGeom &g = r.d;
    const size_t per_image = (size_t)C * g.plane;
    std::vector<float> buf((size_t)g.FS + B * per_image + 3 * g.Wp + 64, 0.f);
    float *xp = buf.data() + g.FS;
    for (int bc = 0; bc < B * C; ++bc)
        for (int yy = 0; yy < H; ++yy)
            memcpy(&xp[((size_t)bc * g.Hp + yy + g.p0) * g.Wp + g.p0], x + ((size_t)bc * H + yy) * W, sizeof(float) * W);
    if (r.recs_per_wave != g.K + 1) return -20;
    // the stream visits every tap exactly once, columns in non-decreasing alignment class
    {
        std::vector<int> seen(g.K, 0);
        int last = -1;
        for (int i = 0; i < g.K; ++i) {
            const int t = ring_stream_tap(r, i);
            if (t < 0 || t >= g.K || seen[t]++) return -21;
            const int sa = (int)((r.cols_sa >> (4 * (i / k))) & 15);
            if (sa != diag_align(g, t % k) || sa < last) return -22;
            last = sa;
        }
    }
    memset(y, 0, sizeof(float) * (size_t)B * C * g.L);
    memset(count, 0, sizeof(int) * (size_t)B * g.L);
    const size_t Lz = (size_t)g.L;
    std::vector<uint32_t> rec(g.RECW), prev(g.RECW);
    std::vector<float> acc((size_t)C * k);
    for (int b = 0; b < B; ++b) {
        const float *base = buf.data() + (size_t)b * per_image;   // FS floats before image b's planes
        const float *im = img + (size_t)b * 3 * Lz;
        for (int beta = 0; beta < g.NBB; ++beta)
            for (int gamma = 0; gamma < g.NG; ++gamma)
                for (int lane = 0; lane < g.BL; ++lane) {           // the lanes that own pixels
                    std::fill(acc.begin(), acc.end(), 0.f);
                    bool any = false;
                    for (int n = 0; n < k; ++n) {
                        const long long l = (long long)gamma * k + n + ((long long)g.BL * beta + lane - n) * g.CK;
                        if ((long long)g.BL * beta + lane - n >= 0 && l < g.L) any = true;
                    }
                    if (!any) continue;
                    for (int i = 0; i < g.K; ++i) {
                        const int t = ring_stream_tap(r, i), ky = t / k, kx = t % k;
                        const int sa = diag_align(g, kx), na = k - sa;
                        for (int ci = 0; ci < g.NCS; ++ci) {
                            ring_fill_record(r, beta, gamma, i, ci, rec.data());
                            // which lanes past BL loaded along for this position: all of them at position 0 (the
                            // loads before the loop), otherwise what the previous record's flag word says
                            uint32_t wide = ~0u;
                            if (i > 0) {
                                ring_fill_record(r, beta, gamma, i - 1, ci, prev.data());
                                wide = prev[2 * g.CH + (g.CH + 1) / 2];
                            }
                            for (int s = 0; s < g.CH; ++s) {
                                const int c = ci * g.CH + s;
                                if (c >= C) continue;
                                const uint32_t t16 = (rec[2 * g.CH + s / 2] >> (16 * (s & 1))) & 0xffffu;
                                if ((t16 != 0x8080u) != (((wide >> s) & 1u) != 0) && i > 0) return -25;   // flag word out of step
                                const int tA = (int)(t16 & 255u) - 64, tB = (int)(t16 >> 8) - 64;
                                for (int n = 0; n < k; ++n) {
                                    const bool useA = n < na;
                                    const int off = (int)(rec[useA ? s : g.CH + s] / 4);
                                    const int T = useA ? tA : tB;
                                    int src_lane = lane;
                                    if (t16 != 0x8080u && lane - T >= n) src_lane = lane + g.gap;   // the lane permute
                                    if (src_lane > 63) return -23;          // would wrap: only 65 - k lanes may own pixels
                                    if (src_lane >= g.BL && !((wide >> s) & 1u)) return -24;   // that lane requested nothing
                                    acc[(size_t)c * k + n] += 0.f;          // (weights applied below, per pixel)
                                    // tap weight of pixel n
                                    const long long bb = (long long)g.BL * beta + lane - n;
                                    const long long l = (long long)gamma * k + n + bb * g.CK;
                                    if (bb < 0 || l >= g.L) continue;
                                    const int py = (int)(l / W), px = (int)(l % W);
                                    const int ny = py + ky - g.p0, nx = px + kx - g.p0;
                                    const bool in = ny >= 0 && ny < H && nx >= 0 && nx < W;
                                    float cd = 0.f;
                                    for (int ch = 0; ch < 3; ++ch) {
                                        const float d = im[ch * Lz + l] - (in ? im[ch * Lz + (size_t)ny * W + nx] : 0.f);
                                        cd += d * d;
                                    }
                                    const int dy = ky - g.p0, dx = kx - g.p0;
                                    const float wgt = (float)(dy * dy + dx * dx) * pw + cd * cw;
                                    acc[(size_t)c * k + n] += wgt * base[off + src_lane];
                                }
                            }
                        }
                    }
                    for (int n = 0; n < k; ++n) {
                        const long long bb = (long long)g.BL * beta + lane - n;
                        const long long l = (long long)gamma * k + n + bb * g.CK;
                        if (bb < 0 || l >= g.L) continue;
                        count[(size_t)b * Lz + l] += 1;
                        for (int c = 0; c < C; ++c)
                            y[((size_t)b * C + c) * Lz + l] = acc[(size_t)c * k + n] * x[((size_t)b * C + c) * Lz + l];
                    }
                }
        // the pixels of a nearly empty last b-block go to tail workgroups: computed here from the definition
        for (long long idx = 0; g.tail_lanes && idx < diag_tail_candidates(g); ++idx) {
            const int l = diag_tail_pixel(g, idx);
            if (l < 0) continue;
            count[(size_t)b * Lz + l] += 1;
            const int py = l / W, px = l % W;
            const float *xpad = buf.data() + g.FS + (size_t)b * per_image;
            for (int c = 0; c < C; ++c) {
                float a = 0.f;
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
                    a += ((float)(dy * dy + dx * dx) * pw + cd * cw) * val;
                }
                y[((size_t)b * C + c) * Lz + l] = a * x[((size_t)b * C + c) * Lz + l];
            }
        }
    }
    return 0;
}

// Geometry figures for the tests: out[0..7] = NBB, NG, NGW, NCS, CH, BL, tail_lanes, RECW.
extern "C" int emu_ring_geom(int B, int C, int H, int W, int k, int *out)
{
    RingGeom r;
    const int rc = make_ring_geom(r, B, C, H, W, k);
    if (rc) return rc;
    const DiagGeom &g = r.d;
    const int v[8] = {g.NBB, g.NG, g.NGW, g.NCS, g.CH, g.BL, g.tail_lanes, g.RECW};
    for (int i = 0; i < 8; ++i) out[i] = v[i];
    return 0;
}
