// Stand-alone experiment (not part of the library or the tests): the GATHER form of the local operator's data
// gradient, timed next to what the library's scatter form costs.
//   scattered term of dx[c'][y][x] = sum over the K window positions (ky', kx') that reach the element of
//       D[c][l] * w[l][t]     with (c, t, l) from  m = l' * CK + c' * K + ky' * k + kx' = (c * K + t) * L + l
//   (LocalConvolution.py:79-90 read backwards; D = dy * x, w = pos_w * pd[t] + col_w * |I(l) - I(l + tap t)|^2).
// Consecutive elements x (the lanes of a wave) have consecutive source pixels l', so m moves by CK per lane:
// l = a + b * CK moves along b.  With D and the weights stored transposed ([a][b] instead of [l]) both loads of a
// term are coalesced; the two transpositions are the preparation passes timed with the kernel.
// Build: hipcc -O3 -std=c++17 --offload-arch=gfx950 exp_dx_gather.hip -o exp_dx_gather
// Run:   ./exp_dx_gather [B C H W k]   (default 8 21 512 512 7); small shapes are checked against a plain scatter.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK_(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

struct G { int B, C, H, W, k, K, L, CK, NB, p0, Hp, Wp; };

// D_T[b][c][a][bb] = D[b][c][a + bb * CK] (0 past the image)
__global__ void transpose_d(G g, const float *__restrict__ D, float *__restrict__ DT)
{
    const size_t n = (size_t)g.B * g.C * g.CK * g.NB;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int bb = (int)(i % g.NB);
        const size_t r = i / g.NB;
        const int a = (int)(r % g.CK);
        const size_t bc = r / g.CK;
        const long long l = a + (long long)bb * g.CK;
        DT[i] = l < g.L ? D[bc * g.L + l] : 0.f;
    }
}

// w_T[b][t][a][bb] = pos_w * pd[t] + col_w * |I(l) - I(l + tap t)|^2, l = a + bb * CK (img: [B][3][H][W], zero outside)
__global__ void transpose_w(G g, const float *__restrict__ img, float pw, float cw, float *__restrict__ WT)
{
    const size_t n = (size_t)g.B * g.K * g.CK * g.NB;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int bb = (int)(i % g.NB);
        size_t r = i / g.NB;
        const int a = (int)(r % g.CK);
        r /= g.CK;
        const int t = (int)(r % g.K), b = (int)(r / g.K);
        const long long l = a + (long long)bb * g.CK;
        float w = 0.f;
        if (l < g.L) {
            const int y = (int)(l / g.W), x = (int)(l - (long long)y * g.W);
            const int ty = t / g.k, tx = t - ty * g.k;
            const int yy = y + ty - g.p0, xx = x + tx - g.p0;
            float cd = 0.f;
            for (int ch = 0; ch < 3; ++ch) {
                const float c0 = img[((size_t)(b * 3 + ch) * g.H + y) * g.W + x];
                const float c1 = (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? img[((size_t)(b * 3 + ch) * g.H + yy) * g.W + xx] : 0.f;
                cd += (c0 - c1) * (c0 - c1);
            }
            const int dy = ty - g.p0, dx = tx - g.p0;
            w = pw * (float)(dy * dy + dx * dx) + cw * cd;
        }
        WT[i] = w;
    }
}

// one thread per element (b, c', y, x), x fastest
__global__ void __launch_bounds__(256)
dx_gather(G g, const float *__restrict__ DT, const float *__restrict__ WT, float *__restrict__ dx)
{
    const size_t n = (size_t)g.B * g.C * g.L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int x = (int)(i % g.W);
    size_t r = i / g.W;
    const int y = (int)(r % g.H);
    r /= g.H;
    const int cp = (int)(r % g.C), b = (int)(r / g.C);
    const float *dt = DT + (size_t)b * g.C * g.CK * g.NB;
    const float *wt = WT + (size_t)b * g.K * g.CK * g.NB;
    const size_t plane = (size_t)g.CK * g.NB;
    float acc = 0.f;
    for (int ky = 0; ky < g.k; ++ky) {
        const int ys = y + g.p0 - ky;
        if (ys < 0 || ys >= g.H) continue;
        // entries kx = kx0 .. kx1 have their source pixel inside the row
        const int kx0 = x + g.p0 - (g.W - 1) > 0 ? x + g.p0 - (g.W - 1) : 0;
        const int kx1 = x + g.p0 < g.k - 1 ? x + g.p0 : g.k - 1;
        if (kx0 > kx1) continue;
        const int xs = x + g.p0 - kx0;
        const unsigned m = (unsigned)(ys * g.W + xs) * (unsigned)g.CK + (unsigned)(cp * g.K + ky * g.k + kx0);
        unsigned q = m / (unsigned)g.L;
        int l = (int)(m - q * (unsigned)g.L);
        int a = l % g.CK, bb = l / g.CK;
        int c = (int)(q / (unsigned)g.K), t = (int)(q - (unsigned)c * g.K);
        for (int kx = kx0; kx <= kx1; ++kx) {
            acc = fmaf(dt[(size_t)c * plane + (size_t)a * g.NB + bb], wt[(size_t)t * plane + (size_t)a * g.NB + bb], acc);
            // the next kx: one source pixel back, one window column on: m -= CK - 1
            l -= g.CK - 1;
            if (l >= 0) {
                if (++a == g.CK) a = 0; else --bb;
            } else {
                l += g.L;
                a = l % g.CK; bb = l / g.CK;
                if (--t < 0) { t = g.K - 1; --c; }
            }
        }
    }
    dx[i] = acc;
}

// reference: plain scatter with float atomics, one thread per (b, c, l), looping over the taps
__global__ void dx_scatter_ref(G g, const float *__restrict__ D, const float *__restrict__ img, float pw, float cw, float *__restrict__ dxp)
{
    const size_t n = (size_t)g.B * g.C * g.L;
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int l = (int)(i % g.L);
    const size_t bc = i / g.L;
    const int c = (int)(bc % g.C), b = (int)(bc / g.C);
    const int y = l / g.W, x = l - y * g.W;
    const float d = D[i];
    for (int t = 0; t < g.K; ++t) {
        const int ty = t / g.k, tx = t - ty * g.k;
        const int yy = y + ty - g.p0, xx = x + tx - g.p0;
        float cd = 0.f;
        for (int ch = 0; ch < 3; ++ch) {
            const float c0 = img[((size_t)(b * 3 + ch) * g.H + y) * g.W + x];
            const float c1 = (yy >= 0 && yy < g.H && xx >= 0 && xx < g.W) ? img[((size_t)(b * 3 + ch) * g.H + yy) * g.W + xx] : 0.f;
            cd += (c0 - c1) * (c0 - c1);
        }
        const int dy = ty - g.p0, dxo = tx - g.p0;
        const float w = pw * (float)(dy * dy + dxo * dxo) + cw * cd;
        const unsigned long long m = ((unsigned long long)c * g.K + t) * (unsigned long long)g.L + l;
        const int lp = (int)(m / g.CK), rr = (int)(m - (unsigned long long)lp * g.CK);
        const int cs = rr / g.K, ts = rr - cs * g.K;
        const int Y = lp / g.W + ts / g.k, X = lp % g.W + ts % g.k;        // padded coordinates
        const int yo = Y - g.p0, xo = X - g.p0;
        if (yo >= 0 && yo < g.H && xo >= 0 && xo < g.W) atomicAdd(&dxp[((size_t)(b * g.C + cs) * g.H + yo) * g.W + xo], w * d);
    }
}

int main(int argc, char **argv)
{
    G g{8, 21, 512, 512, 7};
    if (argc >= 6) { g.B = atoi(argv[1]); g.C = atoi(argv[2]); g.H = atoi(argv[3]); g.W = atoi(argv[4]); g.k = atoi(argv[5]); }
    g.K = g.k * g.k; g.L = g.H * g.W; g.CK = g.C * g.K; g.NB = (g.L + g.CK - 1) / g.CK; g.p0 = g.k / 2;
    const size_t n = (size_t)g.B * g.C * g.L, nimg = (size_t)g.B * 3 * g.L;
    const size_t nDT = (size_t)g.B * g.C * g.CK * g.NB, nWT = (size_t)g.B * g.K * g.CK * g.NB;
    std::vector<float> hD(n), hI(nimg);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto &v : hD) v = rnd();
    for (auto &v : hI) v = rnd();
    float *D, *I, *DT, *WT, *dx, *ref;
    CK_(hipMalloc(&D, n * 4)); CK_(hipMalloc(&I, nimg * 4)); CK_(hipMalloc(&DT, nDT * 4)); CK_(hipMalloc(&WT, nWT * 4));
    CK_(hipMalloc(&dx, n * 4)); CK_(hipMalloc(&ref, n * 4));
    CK_(hipMemcpy(D, hD.data(), n * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(I, hI.data(), nimg * 4, hipMemcpyHostToDevice));
    const float pw = -0.004f, cw = -0.003f;
    hipEvent_t e0, e1, e2, e3;
    CK_(hipEventCreate(&e0)); CK_(hipEventCreate(&e1)); CK_(hipEventCreate(&e2)); CK_(hipEventCreate(&e3));
    std::vector<float> tp, tw, tg;
    for (int it = 0; it < 6; ++it) {
        CK_(hipEventRecord(e0));
        hipLaunchKernelGGL(transpose_d, dim3(8192), dim3(256), 0, 0, g, D, DT);
        CK_(hipEventRecord(e1));
        hipLaunchKernelGGL(transpose_w, dim3(8192), dim3(256), 0, 0, g, I, pw, cw, WT);
        CK_(hipEventRecord(e2));
        hipLaunchKernelGGL(dx_gather, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, g, DT, WT, dx);
        CK_(hipEventRecord(e3));
        CK_(hipEventSynchronize(e3));
        float a, b, c;
        CK_(hipEventElapsedTime(&a, e0, e1)); CK_(hipEventElapsedTime(&b, e1, e2)); CK_(hipEventElapsedTime(&c, e2, e3));
        if (it) { tp.push_back(a); tw.push_back(b); tg.push_back(c); }
    }
    auto med = [](std::vector<float> v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; };
    printf("B%d C%d %dx%d k%d: transpose D %.3f ms, weight table (transposed, %.0f MB) %.3f ms, gather %.3f ms\n", g.B, g.C, g.H, g.W, g.k,
           med(tp), nWT * 4 / 1e6, med(tw), med(tg));
    if (n <= (size_t)4 << 20) {
        CK_(hipMemset(ref, 0, n * 4));
        hipLaunchKernelGGL(dx_scatter_ref, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, g, D, I, pw, cw, ref);
        std::vector<float> a(n), b(n);
        CK_(hipMemcpy(a.data(), dx, n * 4, hipMemcpyDeviceToHost));
        CK_(hipMemcpy(b.data(), ref, n * 4, hipMemcpyDeviceToHost));
        double worst = 0, scale = 0;
        for (size_t i = 0; i < n; ++i) { worst = std::max(worst, (double)fabsf(a[i] - b[i])); scale = std::max(scale, (double)fabsf(b[i])); }
        printf("check against the plain scatter: max |diff| %.3e at scale %.3e\n", worst, scale);
    }
    return 0;
}
