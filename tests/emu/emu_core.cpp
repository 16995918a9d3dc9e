// Host emulation of the gather core (semantic_segmentation_amd/csrc/mfl_core.h): every GPU
// thread of local_forward_kernel is run as a loop iteration on the CPU, with the same
// geometry, offset table, column mapping and incremental decode.  It lets the CPU test
// suite check the index arithmetic of the device code against the oracle without a GPU.
// Build: clang++ -O2 -shared -fPIC (the core uses clang vector extensions).
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../../semantic_segmentation_amd/csrc/mfl_core.h"

using namespace mfl;

namespace {

template <int CC, int M, bool FROM_CD>
void run(const Geom &g, const float *xp, const F4 *ip4, const float *cd, const int *tab, float pw, float cw, float *y)
{
    const int nchunks = (g.C + CC - 1) / CC;
    const int nthreads = blocks_per_image(g) * 256;     // include the inactive tail lanes
    const int ctr_off = g.p0 * g.Wp + g.p0;
    for (int b = 0; b < g.B; ++b)
        for (int chunk = 0; chunk < nchunks; ++chunk)
            for (int uraw = 0; uraw < nthreads; ++uraw) {
                const int c0 = chunk * CC, nch = (g.C - c0 < CC) ? g.C - c0 : CC;
                Columns<M> col;
                make_columns<M>(col, g, uraw / 256, uraw % 256);
                const float *xp_b = xp + (size_t)b * g.C * g.plane;
                float acc[CC][M];
                for (int cc = 0; cc < CC; ++cc)
                    for (int n = 0; n < M; ++n) acc[cc][n] = 0.f;
                NoSink sink;
                window_sums<CC, M, FROM_CD, false>(acc, g, xp_b, FROM_CD ? nullptr : ip4 + (size_t)b * g.plane,
                                            FROM_CD ? cd + (size_t)b * g.L * g.K : nullptr, tab, pw, cw, c0, nch, col,
                                            nullptr, sink);
                for (int n = 0; n < M; ++n)
                    if (col.valid[n])
                        for (int cc = 0; cc < nch; ++cc)
                            y[((size_t)b * g.C + c0 + cc) * g.L + col.lcl[n]] =
                                acc[cc][n] * xp_b[(size_t)(c0 + cc) * g.plane + col.ibase[n] + ctr_off];
            }
}

}  // namespace

extern "C" int emu_choose_m(long long L, long long CK, int max_m) { return choose_m(L, CK, max_m); }

// Returns 0 on success; -1/-3 from make_geom; -2 for an (M, CC) pair that is not built here.
extern "C" int emu_local_forward(const float *x, const float *img, const float *cd, float pw, float cw, float *y, int B,
                                 int C, int H, int W, int k, int M, int CC)
{
    Geom g;
    int rc = make_geom(g, B, C, H, W, k, M);
    if (rc) return rc;
    const size_t npad = (size_t)B * C * g.plane + padded_slack(g);
    std::vector<float> xp(npad, 0.f);
    for (int bc = 0; bc < B * C; ++bc)
        for (int yy = 0; yy < H; ++yy)
            memcpy(&xp[((size_t)bc * g.Hp + yy + g.p0) * g.Wp + g.p0], x + ((size_t)bc * H + yy) * W, sizeof(float) * W);
    std::vector<F4> ip4((size_t)B * g.plane + 64, F4{0, 0, 0, 0});
    if (img)
        for (int b = 0; b < B; ++b)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    const float *p = img + (size_t)b * 3 * g.L + (size_t)yy * W + xx;
                    ip4[(size_t)b * g.plane + (size_t)(yy + g.p0) * g.Wp + xx + g.p0] = F4{p[0], p[g.L], p[2 * (size_t)g.L], 0.f};
                }
    std::vector<int> tab(g.CK);
    for (int r = 0; r < g.CK; ++r) tab[r] = table_entry(g, r);
    memset(y, 0, sizeof(float) * (size_t)B * C * g.L);

#define RUN(CCV, MV)                                                                                          \
    if (CC == CCV && M == MV) {                                                                               \
        if (cd) run<CCV, MV, true>(g, xp.data(), ip4.data(), cd, tab.data(), pw, cw, y);                      \
        else run<CCV, MV, false>(g, xp.data(), ip4.data(), cd, tab.data(), pw, cw, y);                        \
        return 0;                                                                                             \
    }
    RUN(8, 1) RUN(8, 2) RUN(8, 4) RUN(21, 1) RUN(21, 2) RUN(21, 4) RUN(3, 1) RUN(3, 4)
#undef RUN
    return -2;
}
