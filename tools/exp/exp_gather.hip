// Stand-alone timing harness for gather-kernel experiments (not part of the library or the tests).
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -DCCV=21 -DMV=4 exp_gather.hip -o exp_gather
// Run:   ./exp_gather [B C H W k]      prints the median kernel time and a checksum.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../semantic_segmentation_amd/csrc/mfl_core.h"

using namespace mfl;

#ifndef CCV
#define CCV 21
#endif
#ifndef MV
#define MV 4
#endif
#ifndef MINWAVES
#define MINWAVES 1
#endif

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int CC, int M, bool EXACT>
__global__ void __launch_bounds__(256, MINWAVES)
fwd_kernel(Geom g, const float *__restrict__ xp, const F4 *__restrict__ ip4, const int *__restrict__ tab_g,
           float pw, float cw, float *__restrict__ y)
{
    extern __shared__ __attribute__((aligned(16))) int lds_tab[];
    for (int r = threadIdx.x; r < g.CK; r += 256) lds_tab[r] = tab_g[r];
    __syncthreads();
    const int b = blockIdx.z, c0 = blockIdx.y * CC;
    const int nch = min(CC, g.C - c0);
    Columns<M> col;
    make_columns<M>(col, g, blockIdx.x, threadIdx.x);
    const float *xp_b = xp + (size_t)b * g.C * g.plane;
    const F4 *ip_b = ip4 + (size_t)b * g.plane;
    float acc[CC][M];
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
#pragma unroll
        for (int n = 0; n < M; ++n) acc[cc][n] = 0.f;
    NoSink sink;
    window_sums<CC, M, false, EXACT>(acc, g, xp_b, ip_b, nullptr, lds_tab, pw, cw, c0, nch, col, nullptr, sink);
    const int ctr_off = g.p0 * g.Wp + g.p0;
    float *y_b = y + (size_t)b * g.C * g.L;
#pragma unroll
    for (int n = 0; n < M; ++n)
        if (col.valid[n])
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
                if (cc < nch)
                    y_b[(size_t)(c0 + cc) * g.L + col.lcl[n]] =
                        acc[cc][n] * xp_b[(size_t)(c0 + cc) * g.plane + col.ibase[n] + ctr_off];
}

int main(int argc, char **argv)
{
    int B = 8, C = 21, H = 512, W = 512, k = 7;
    if (argc >= 6) { B = atoi(argv[1]); C = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); k = atoi(argv[5]); }
    Geom g;
    if (make_geom(g, B, C, H, W, k, MV)) { printf("bad geom\n"); return 1; }
    const size_t npad = (size_t)B * C * g.plane + padded_slack(g);
    std::vector<float> hx(npad, 0.f);
    std::vector<F4> hi((size_t)B * g.plane + 64, F4{0, 0, 0, 0});
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 32768.f - 1.f; };
    for (int bc = 0; bc < B * C; ++bc)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx) hx[((size_t)bc * g.Hp + yy + g.p0) * g.Wp + xx + g.p0] = rnd();
    for (int b = 0; b < B; ++b)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx)
                hi[(size_t)b * g.plane + (size_t)(yy + g.p0) * g.Wp + xx + g.p0] = F4{rnd(), rnd(), rnd(), 0.f};
    std::vector<int> ht(g.CK);
    for (int r = 0; r < g.CK; ++r) ht[r] = table_entry(g, r);
    float *xp, *y; F4 *ip4; int *tab;
    CK_(hipMalloc(&xp, npad * 4)); CK_(hipMalloc(&ip4, hi.size() * sizeof(F4))); CK_(hipMalloc(&tab, g.CK * 4));
    CK_(hipMalloc(&y, (size_t)B * C * g.L * 4));
    CK_(hipMemcpy(xp, hx.data(), npad * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(ip4, hi.data(), hi.size() * sizeof(F4), hipMemcpyHostToDevice));
    CK_(hipMemcpy(tab, ht.data(), g.CK * 4, hipMemcpyHostToDevice));
    const dim3 grid(blocks_per_image(g), (C + CCV - 1) / CCV, B);
    auto launch = [&]() {
        if (C % CCV == 0)
            hipLaunchKernelGGL((fwd_kernel<CCV, MV, true>), grid, dim3(256), g.CK * 4, 0, g, xp, ip4, tab, -0.004f, -0.003f, y);
        else
            hipLaunchKernelGGL((fwd_kernel<CCV, MV, false>), grid, dim3(256), g.CK * 4, 0, g, xp, ip4, tab, -0.004f, -0.003f, y);
    };
    launch(); CK_(hipDeviceSynchronize());
    std::vector<float> ms;
    hipEvent_t e0, e1; CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1));
    for (int i = 0; i < 7; ++i) {
        CK_(hipEventRecord(e0)); launch(); CK_(hipEventRecord(e1)); CK_(hipEventSynchronize(e1));
        float t; CK_(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
    }
    std::sort(ms.begin(), ms.end());
    std::vector<float> hy((size_t)B * C * g.L);
    CK_(hipMemcpy(hy.data(), y, hy.size() * 4, hipMemcpyDeviceToHost));
    double sum = 0, asum = 0;
    for (float v : hy) { sum += v; asum += v < 0 ? -v : v; }
    const double elems = (double)B * C * g.K * g.L;
    printf("CC=%d M=%d tiles/wave=%d grid=(%d,%d,%d) median %.4f ms  min %.4f ms  %.2f Gelem/s  %.1f Mpix/s  checksum %.6e %.6e\n", CCV, MV,
           kWaveTiles, grid.x, grid.y, grid.z, ms[ms.size() / 2], ms[0], elems / ms[ms.size() / 2] / 1e6,
           (double)B * g.L / ms[ms.size() / 2] / 1e3, sum, asum);
    return 0;
}
