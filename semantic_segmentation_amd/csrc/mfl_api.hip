// HIP kernels (gfx950) and the C ABI of the local-window message-passing library.
// Declarations and the reference symbols each entry point replaces: include/mfl.h.
//
// Data layout in device memory
//   x / unary / y / logq      fp32 NCHW as the caller holds them
//   xp  (workspace)           zero-padded planes [B][C][Hp][Wp] (+ slack), Hp = H+k-1, Wp = W+k-1:
//                             the gather never needs a bounds check, and the mean-field loop writes
//                             Q_{t+1} straight into the interior of the next padded buffer
//   ip4 (workspace)           zero-padded RGB image, one 16-byte {r,g,b,0} per pixel [B][Hp][Wp]:
//                             one 16-byte load per (pixel, tap) for the colour distance
//   tab (workspace)           CK int32 offsets, staged into LDS by every block
//
// Thread mapping, index map and the incremental decode: mfl_core.h; gather kernels: mfl_gather.hip.
#include <hip/hip_runtime.h>
#include <atomic>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/mfl.h"
#include "mfl_gather.h"
#include "mfl_diag_api.h"
#include "mfl_ring_api.h"

using namespace mfl;

namespace {

constexpr int kMaxLdsBytes = 60 * 1024;  // dynamic LDS one block may ask for without raising the launch limit

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

#define MFL_CHECK_HIP(expr)                          \
    do {                                             \
        hipError_t e_ = (expr);                      \
        if (e_ != hipSuccess) return (int)e_;        \
    } while (0)

// ------------------------------------------------------------------ small kernels

// dst[bc][yy][xx] = src[bc][yy-p0][xx-p0] inside the image, 0 on the border.
__global__ void pad_planes_kernel(const float *__restrict__ src, float *__restrict__ dst, int planes, int H, int W,
                                  int Hp, int Wp, int p0)
{
    // one padded row per workgroup pass: a single division per row, coalesced along x
    const size_t rows = (size_t)planes * Hp;
    for (size_t row = blockIdx.x; row < rows; row += gridDim.x) {
        const size_t pc = row / Hp;
        const int y = (int)(row - pc * Hp) - p0;
        const bool inside = y >= 0 && y < H;
        const float *s = src + (pc * H + (inside ? y : 0)) * W;
        float *d = dst + row * Wp;
        for (int xx = threadIdx.x; xx < Wp; xx += blockDim.x)
            d[xx] = (inside && xx >= p0 && xx < W + p0) ? s[xx - p0] : 0.0f;
    }
}

// dst[b][yy][xx] = {r, g, b, 0} of the padded image.
__global__ void pad_image4_kernel(const float *__restrict__ img, F4 *__restrict__ dst, int B, int H, int W, int Hp,
                                  int Wp, int p0)
{
    const size_t n = (size_t)B * Hp * Wp;
    const size_t L = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % Wp);
        const size_t rest = i / Wp;
        const int yy = (int)(rest % Hp);
        const size_t b = rest / Hp;
        const int y = yy - p0, x = xx - p0;
        F4 v = {0.f, 0.f, 0.f, 0.f};
        if (y >= 0 && y < H && x >= 0 && x < W) {
            const float *p = img + b * 3 * L + (size_t)y * W + x;
            v.x = p[0]; v.y = p[L]; v.z = p[2 * L];
        }
        dst[i] = v;
    }
}

__global__ void build_table_kernel(int *__restrict__ tab, Geom g)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < g.CK) tab[r] = table_entry(g, r);
}

// cd[b][l][t]: one thread per output element, consecutive lanes write consecutive taps.
__global__ void color_distance_kernel(const float *__restrict__ img, float *__restrict__ cd, int B, int H, int W, int k)
{
    const int K = k * k, p0 = k / 2;
    const size_t L = (size_t)H * W, n = (size_t)B * L * K;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int t = (int)(i % K);
        const size_t bl = i / K;
        const size_t l = bl % L, b = bl / L;
        const int y = (int)(l / W), x = (int)(l % W);
        const int ny = y + t / k - p0, nx = x + t % k - p0;
        const bool in = ny >= 0 && ny < H && nx >= 0 && nx < W;
        const float *p = img + b * 3 * L;
        float s = 0.f;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) {
            const float d = p[ch * L + l] - (in ? p[ch * L + (size_t)ny * W + nx] : 0.f);
            s += d * d;
        }
        cd[i] = s;
    }
}

// logq = LogSoftmax(log_softmax(unary) - min_m smooth[m*C + c]); one thread per pixel, runtime C.
// q (optional) = exp(logq), written either unpadded ([B][C][H][W], q_pad_k = 0) or into the
// interior of zero-padded planes for a k x k window (q_pad_k = k).
__global__ void combine_kernel(const float *__restrict__ unary, const float *__restrict__ smooth, int mixtures,
                               float *__restrict__ logq, float *__restrict__ q, int q_pad_k, int B, int C, int H,
                               int W)
{
    const size_t L = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (size_t)B * L) return;
    const size_t b = pix / L, l = pix % L;
    const float *u = unary + b * C * L + l;
    const float *s = smooth + b * mixtures * C * L + l;

    float mu = -INFINITY;
    for (int c = 0; c < C; ++c) mu = fmaxf(mu, u[c * L]);
    float su = 0.f;
    for (int c = 0; c < C; ++c) su += expf(u[c * L] - mu);
    const float lse_u = mu + logf(su);

    auto v_at = [&](int c) {
        float mn = s[c * L];
        for (int m = 1; m < mixtures; ++m) mn = fminf(mn, s[((size_t)m * C + c) * L]);
        return (u[c * L] - lse_u) - mn;
    };
    float mv = -INFINITY;
    for (int c = 0; c < C; ++c) mv = fmaxf(mv, v_at(c));
    float sv = 0.f;
    for (int c = 0; c < C; ++c) sv += expf(v_at(c) - mv);
    const float lse_v = mv + logf(sv);

    size_t qoff = 0, qplane = L;
    if (q) {
        if (q_pad_k > 0) {
            const int Wp = W + q_pad_k - 1, Hp = H + q_pad_k - 1, p0 = q_pad_k / 2;
            qplane = (size_t)Hp * Wp;
            qoff = b * C * qplane + (size_t)((int)(l / W) + p0) * Wp + (int)(l % W) + p0;
        } else {
            qoff = b * C * L + l;
        }
    }
    for (int c = 0; c < C; ++c) {
        const float lq = v_at(c) - lse_v;
        if (logq) logq[b * C * L + c * L + l] = lq;
        if (q) q[qoff + c * qplane] = expf(lq);
    }
}

// msg <- compat . (aw * msg + ab), per pixel, runtime C (the large-label path; the fused
// kernel does this in registers for C <= 32).  One block handles 64 pixels: the 64 x C tile of
// messages is staged in LDS, the C x C matrix is read through the cache.
__global__ void label_transform_kernel(const float *msg, const float *__restrict__ aw,
                                       const float *__restrict__ ab, const float *__restrict__ compat,
                                       float *out, int B, int C, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];  // [C][64]
    const size_t L = (size_t)H * W;
    const int px = threadIdx.x & 63, grp = threadIdx.x >> 6;       // 4 groups of 64 lanes
    const size_t pix = (size_t)blockIdx.x * 64 + px;
    const bool ok = pix < (size_t)B * L;
    const size_t b = ok ? pix / L : 0, l = ok ? pix % L : 0;
    for (int c = grp; c < C; c += 4) {
        float v = ok ? msg[(b * C + c) * L + l] : 0.f;
        if (aw) v = fmaf(aw[(size_t)c * L + l], v, ab[(size_t)c * L + l]);
        tile[c * 64 + px] = v;
    }
    __syncthreads();
    for (int c = grp; c < C; c += 4) {
        float acc;
        if (compat) {
            acc = 0.f;
            const float *row = compat + (size_t)c * C;
            for (int c2 = 0; c2 < C; ++c2) acc = fmaf(row[c2], tile[c2 * 64 + px], acc);
        } else {
            acc = tile[c * 64 + px];
        }
        if (ok) out[(b * C + c) * L + l] = acc;
    }
}

__global__ void argmax_kernel(const float *__restrict__ logq, unsigned char *__restrict__ labels, int B, int C, int H,
                              int W)
{
    const size_t L = (size_t)H * W;
    const size_t pix = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= (size_t)B * L) return;
    const size_t b = pix / L, l = pix % L;
    const float *p = logq + b * C * L + l;
    float best = p[0];
    int arg = 0;
    for (int c = 1; c < C; ++c) {
        const float v = p[c * L];
        if (v > best) { best = v; arg = c; }
    }
    labels[pix] = (unsigned char)arg;
}

// counts[0][c] += #(pred == c and target == c), counts[1][c] += #(pred == c), counts[2][c] += #(target == c):
// a histogram per workgroup in LDS, then one 64-bit integer atomic per workgroup and counter (integer
// sums: the result does not depend on the order).  Labels >= C (e.g. 255 = void) fall in no class.
__global__ void __launch_bounds__(kThreads)
iou_counts_kernel(const unsigned char *__restrict__ pred, const unsigned char *__restrict__ target,
                  unsigned long long *__restrict__ counts, size_t n, int C)
{
    extern __shared__ unsigned int hist[];       // [3][C]
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x) hist[i] = 0;
    __syncthreads();
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int p = pred[i], t = target[i];
        if (p < C) atomicAdd(&hist[C + p], 1u);
        if (t < C) {
            atomicAdd(&hist[2 * C + t], 1u);
            if (p == t) atomicAdd(&hist[t], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * C; i += blockDim.x)
        if (hist[i]) atomicAdd(&counts[i], (unsigned long long)hist[i]);
}

// dx = dxd + interior(dxp)
__global__ void backward_finish_kernel(const float *__restrict__ dxd, const float *__restrict__ dxp,
                                       float *__restrict__ dx, int planes, int H, int W, int Hp, int Wp, int p0)
{
    const size_t n = (size_t)planes * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W);
        const size_t rest = i / W;
        const int y = (int)(rest % H);
        const size_t pc = rest / H;
        dx[i] = dxd[i] + dxp[(pc * Hp + y + p0) * Wp + x + p0];
    }
}

// dparams[0..1] = sum over blocks of partials, one block, fixed order (deterministic), in double.
__global__ void reduce_partials_kernel(const float *__restrict__ partials, int nblocks, float *__restrict__ dparams)
{
    __shared__ double sh[2][kThreads];
    double a = 0.0, c = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += kThreads) { a += partials[2 * i]; c += partials[2 * i + 1]; }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { dparams[0] = (float)sh[0][0]; dparams[1] = (float)sh[1][0]; }
}

// ------------------------------------------------------------------ host side

int pad_rows_grid(size_t rows) { return (int)(rows < 65535u * 16u ? rows : 65535u * 16u); }
int grid_for(size_t n) { return (int)((n + kThreads - 1) / kThreads < 65535 * 16 ? (n + kThreads - 1) / kThreads : 65535 * 16); }

// Channel-chunk widths the gather kernels are built for (mfl_gather.hip, -DMFL_CC=...).
int pick_cc(int C)
{
    const int widths[] = {8, 19, 21, 32};
    for (int w : widths)
        if (C <= w) return w;
    return 32;
}

int pick_m(const Geom &g1, int cc, int max_m)
{
    int m = choose_m(g1.L, g1.CK, max_m);
    while (m > 1 && cc * m > 96) m >>= 1;   // accumulator tile within the register budget (kMaxM in mfl_gather.hip)
    return m;
}

struct Workspace {
    // carve-up of the caller's buffer; all offsets 256-byte aligned
    size_t xp, xp2, ip4, tab, dxd, dxp, partials, msg, total;
    size_t psum, qsum, dxp64, consts, partials2;     // tap-split backward only
};

constexpr int kDiagBackwardBlocks = 2048;    // blocks (and partial sums) of the backward's finishing pass

size_t padded_floats(const Geom &g) { return (size_t)g.B * g.C * g.plane + padded_slack(g); }

int backward_blocks(const Geom &g, int cc)
{
    return blocks_per_image(g) * ((g.C + cc - 1) / cc) * g.B;
}

// ---- plan tag ----------------------------------------------------------------------------------
// The first 256 bytes of a workspace identify what the library last laid out in it: {magic, kernel family,
// operation, B, C, H, W, k}.  The address tables depend on nothing else, so a call that finds its own tag
// skips rebuilding them (the table kernels compare the tag on the device and return; no host read-back, no
// state in the library).  Every call that uses the workspace writes its tag after its tables, so a call of
// another operation or size invalidates the plan; a caller that writes to the workspace itself must clear
// these bytes.
struct PlanTag {
    uint32_t v[8];
};

PlanTag make_tag(int family, int op, int B, int C, int H, int W, int k)
{
    return PlanTag{{0x4d464c31u, (uint32_t)family, (uint32_t)op, (uint32_t)B, (uint32_t)C, (uint32_t)H, (uint32_t)W, (uint32_t)k}};
}

__device__ __forceinline__ bool tag_matches(const uint32_t *__restrict__ tag, const PlanTag &want)
{
    bool same = true;
#pragma unroll
    for (int i = 0; i < 8; ++i) same = same && tag[i] == want.v[i];
    return same;
}

__global__ void stamp_tag_kernel(uint32_t *tag, PlanTag t)
{
    if (threadIdx.x < 8) tag[threadIdx.x] = t.v[threadIdx.x];
}

constexpr size_t kTagBytes = 256;

Workspace carve(int op, const Geom &g, int cc)
{
    Workspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    take(kTagBytes);                                  // plan tag
    w.xp = take(padded_floats(g) * 4);
    w.ip4 = take(((size_t)g.B * g.plane + 64) * sizeof(F4));
    w.tab = take((size_t)g.CK * 4);
    if (op == MFL_OP_LOCAL_BACKWARD) {
        w.dxd = take((size_t)g.B * g.C * g.L * 4);
        w.dxp = take(padded_floats(g) * 4);
        w.partials = take((size_t)backward_blocks(g, cc) * 2 * 4);
        if (taps_kernel_applies(g)) {
            // (with an image: the tap-split backward; dxd / dxp above serve the cd-tensor variant of the same shape)
            w.psum = take((size_t)g.B * g.C * g.L * 4);
            w.qsum = take((size_t)g.B * g.C * g.L * 4);
            w.dxp64 = take(padded_floats(g) * 8);
            w.consts = take(8 * 4);
            w.partials2 = take((size_t)2 * kDiagBackwardBlocks * 4);
        }
    }
    if (op == MFL_OP_MEANFIELD_FORWARD) {
        w.xp2 = take(padded_floats(g) * 4);
        if (g.C > 32 || taps_kernel_applies(g)) w.msg = take((size_t)g.B * g.C * g.L * 4);
    }
    w.total = off;
    return w;
}

int check_common(const void *a, const void *b, int B, int C, int H, int W, int k)
{
    if (!a || !b) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0) return MFL_ERR_INVALID_ARGUMENT;
    return MFL_OK;
}

int launch_prepare(void *workspace, const Geom &g, const float *x, const float *img, float *xp, F4 *ip4, int *tab, hipStream_t st)
{
    // (this family rebuilds its small offset table every call; its tag only invalidates other plans)
    hipLaunchKernelGGL(stamp_tag_kernel, dim3(1), dim3(64), 0, st, (uint32_t *)workspace, make_tag(3, 0, g.B, g.C, g.H, g.W, g.k));
    if (x) {
        const size_t n = (size_t)g.B * g.C * g.plane;
        hipLaunchKernelGGL(pad_planes_kernel, dim3(pad_rows_grid((size_t)g.B * g.C * g.Hp)), dim3(kThreads), 0, st, x, xp, g.B * g.C, g.H, g.W,
                           g.Hp, g.Wp, g.p0);
        MFL_CHECK_HIP(hipMemsetAsync(xp + n, 0, padded_slack(g) * 4, st));
    }
    if (img) {
        const size_t n = (size_t)g.B * g.plane;
        hipLaunchKernelGGL(pad_image4_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st, img, ip4, g.B, g.H, g.W, g.Hp,
                           g.Wp, g.p0);
    }
    hipLaunchKernelGGL(build_table_kernel, dim3((g.CK + kThreads - 1) / kThreads), dim3(kThreads), 0, st, tab, g);
    return (int)hipGetLastError();
}

// ---- dispatch over the channel-chunk widths -----------------------------------
#if defined(MFL_DEV_CC)
// development build: only one channel-chunk width (and possibly one kernel kind) is linked
#define MFL_CAT2(a, b) a##b
#define MFL_CAT(a, b) MFL_CAT2(a, b)
#define MFL_BY_CC(cc, fn, ...) \
    if (cc == MFL_DEV_CC) return MFL_CAT(fn##_cc, MFL_DEV_CC)(__VA_ARGS__); \
    return MFL_ERR_INVALID_ARGUMENT;
#else
#define MFL_BY_CC(cc, fn, ...)                    \
    switch (cc) {                                 \
    case 8: return fn##_cc8(__VA_ARGS__);         \
    case 19: return fn##_cc19(__VA_ARGS__);       \
    case 21: return fn##_cc21(__VA_ARGS__);       \
    default: return fn##_cc32(__VA_ARGS__);       \
    }
#endif

#if defined(MFL_DEV_CC) && defined(MFL_DEV_STEP_ONLY)
int dispatch_local_forward(int, const GatherArgs &, float *, hipStream_t) { return MFL_ERR_INVALID_ARGUMENT; }
int dispatch_local_backward(int, const GatherArgs &, const BackwardArgs &, hipStream_t) { return MFL_ERR_INVALID_ARGUMENT; }
#else
int dispatch_local_forward(int cc, const GatherArgs &a, float *y, hipStream_t st)
{
    MFL_BY_CC(cc, launch_local_forward, a, y, st)
}
int dispatch_local_backward(int cc, const GatherArgs &a, const BackwardArgs &b, hipStream_t st)
{
    MFL_BY_CC(cc, launch_local_backward, a, b, st)
}
#endif
#if defined(MFL_DEV_CC) && defined(MFL_DEV_STEP_ONLY)
int dispatch_local_backward_taps(int, const GatherArgs &, const TapsBackwardArgs &, hipStream_t) { return MFL_ERR_INVALID_ARGUMENT; }
#else
int dispatch_local_backward_taps(int cc, const GatherArgs &a, const TapsBackwardArgs &t, hipStream_t st)
{
    MFL_BY_CC(cc, launch_local_backward_taps, a, t, st)
}
#endif
int dispatch_meanfield_step(int cc, const GatherArgs &a, const StepArgs &s, hipStream_t st)
{
    MFL_BY_CC(cc, launch_meanfield_step, a, s, st)
}

// Problem set-up shared by the gather entry points: geometry with the M the dispatcher will use.
int setup(Geom &g, int &cc, bool &tab_lds, int B, int C, int H, int W, int k, bool from_cd, int max_m,
          size_t extra_lds)
{
    Geom g1;
    int rc = make_geom(g1, B, C, H, W, k, 1);
    if (rc) return rc == -3 ? MFL_ERR_TOO_LARGE : MFL_ERR_INVALID_ARGUMENT;
    cc = pick_cc(C);
    tab_lds = (size_t)g1.CK * 4 + 16 + extra_lds <= (size_t)kMaxLdsBytes;
#ifdef MFL_DEV_MAX_M
    max_m = MFL_DEV_MAX_M;   // kernel experiments: force the pixels-per-thread variant
#endif
    int m = (from_cd || !tab_lds) ? 1 : pick_m(g1, cc, max_m);
    rc = make_geom(g, B, C, H, W, k, m);
    if (rc) return rc == -3 ? MFL_ERR_TOO_LARGE : MFL_ERR_INVALID_ARGUMENT;
    return MFL_OK;
}


#ifdef MFL_STAMPS
unsigned long long *g_stamps = nullptr;   // diagnostic builds only: device buffer for in-kernel cycle stamps
#define MFL_SET_DBG(da) (da).dbg = g_stamps
#else
#define MFL_SET_DBG(da) (da).dbg = nullptr
#endif

// ---- diagonal gather path (mfl_diag.hip): images at least 64 + k wide, C <= 33, odd k <= 11 ----
constexpr int kDiagWeightLds = 120 * 1024;   // LDS budget of the double-buffered tap-weight table

// Address table and weight-group list of the diagonal kernels, built on the device.
__global__ void diag_table_kernel(DiagGeom g, uint32_t *__restrict__ table, const uint32_t *__restrict__ tag, PlanTag want)
{
    if (tag_matches(tag, want)) return;               // the workspace holds this plan already
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

__global__ void diag_groups_kernel(DiagGeom g, uint32_t *__restrict__ groups, const uint32_t *__restrict__ tag, PlanTag want)
{
    if (tag_matches(tag, want)) return;
    if (blockIdx.x == 0 && threadIdx.x == 0) diag_groups(g, groups);
}


int launch_diag_table(const DiagGeom &g, uint32_t *table, uint32_t *groups, uint32_t *tag, const PlanTag &want, hipStream_t st)
{
    hipLaunchKernelGGL(diag_groups_kernel, dim3(1), dim3(64), 0, st, g, groups, tag, want);
    const size_t nrec = (size_t)g.NBB * g.NG * g.K * g.NCS;
    const unsigned blocks = (unsigned)((nrec + 255) / 256 < 65535u * 8 ? (nrec + 255) / 256 : 65535u * 8);
    hipLaunchKernelGGL(diag_table_kernel, dim3(blocks), dim3(256), 0, st, g, table, tag, want);
    hipLaunchKernelGGL(stamp_tag_kernel, dim3(1), dim3(64), 0, st, tag, want);
    return (int)hipGetLastError();
}

struct DiagWorkspace {
    size_t xp, xp2, ip4, table, groups, compat, total;
    size_t psum, qsum, consts, partials;     // backward only
};

// c[0..3] = the weight pairs (1, 0), (0, 1) of the two plain window sums; c[6], c[7] = 0 (maxima, as bits)
__global__ void diag_backward_consts_kernel(float *c)
{
    if (threadIdx.x == 0) { c[0] = 1.f; c[1] = 0.f; c[2] = 0.f; c[3] = 1.f; c[6] = 0.f; c[7] = 0.f; }
}

// c[6] = max |dy * x|, c[7] = max |img| (bit patterns of non-negative floats order like unsigned integers;
// a maximum does not depend on the order it is taken in).  NaNs count as the largest value.
__global__ void __launch_bounds__(kThreads)
diag_backward_absmax_kernel(const float *__restrict__ dy, const float *__restrict__ x, size_t n, const float *__restrict__ img,
                            size_t nimg, float *c)
{
    __shared__ unsigned sh[2][kThreads];
    unsigned md = 0, mi = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned b = __float_as_uint(dy[i] * x[i]) & 0x7fffffffu;
        md = b > md ? b : md;
    }
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nimg; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned b = __float_as_uint(img[i]) & 0x7fffffffu;
        mi = b > mi ? b : mi;
    }
    sh[0][threadIdx.x] = md;
    sh[1][threadIdx.x] = mi;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] = sh[0][threadIdx.x] > sh[0][threadIdx.x + s] ? sh[0][threadIdx.x] : sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] = sh[1][threadIdx.x] > sh[1][threadIdx.x + s] ? sh[1][threadIdx.x] : sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        atomicMax(reinterpret_cast<unsigned *>(c) + 6, sh[0][0]);
        atomicMax(reinterpret_cast<unsigned *>(c) + 7, sh[1][0]);
    }
}

// c[4] = scale, c[5] = 1 / scale of the fixed-point scatter: the power of two that keeps every contribution
// |w * D| * scale below 2^50, with |w| <= |pos_w| * max pd + |col_w| * 3 * (2 max|img|)^2.
__global__ void diag_backward_scale_kernel(const float *__restrict__ pos_w, const float *__restrict__ col_w, int k, float *c)
{
    if (threadIdx.x != 0) return;
    const float pdmax = 2.f * (float)((k / 2) * (k / 2));
    const float wmax = fabsf(*pos_w) * pdmax + fabsf(*col_w) * 12.f * c[7] * c[7];
    const float bound = wmax * c[6];
    int ex = 0;
    if (bound > 0.f && bound < INFINITY) frexpf(bound, &ex);       // bound < 2^ex
    int e = 50 - ex;
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
    c[4] = ldexpf(1.f, e);
    c[5] = ldexpf(1.f, -e);
    // a non-finite gradient, input or image (the maxima count NaN as the largest value) has no fixed-point scale:
    // nothing is scattered (scale 0) and every dx comes out NaN, as the reference's autograd would propagate it
    if (!(bound < INFINITY)) { c[4] = 0.f; c[5] = NAN; }
}

// dx = dy * (pos_w * P + col_w * Q) + scattered term (interior of the padded planes of fixed-point sums, consts[5] = 1 / scale);
// partials[2*block + {0,1}] = sum of dy * x * {P, Q} (gradients of position_weight / color_weight),
// where P / Q are the window sums with weights pd[t] / cd[l,t] alone (LocalConvolution.py:84-90).
__global__ void __launch_bounds__(kThreads)
diag_backward_finish_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ P,
                            const float *__restrict__ Q, const long long *__restrict__ dxp, const float *__restrict__ consts,
                            const float *__restrict__ pos_w,
                            const float *__restrict__ col_w, float *__restrict__ dx, float *__restrict__ partials,
                            int planes, int H, int W, int Hp, int Wp, int p0)
{
    __shared__ float sh[2][kThreads];
    const float pw = *pos_w, cw = *col_w;
    const double inv_scale = (double)consts[5];
    const size_t n = (size_t)planes * H * W;
    float a = 0.f, c = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        const size_t rest = i / W;
        const int yy = (int)(rest % H);
        const size_t pc = rest / H;
        const float p = P[i], q = Q[i], g = dy[i], d = g * x[i];
        dx[i] = fmaf(g, fmaf(pw, p, cw * q), (float)((double)dxp[(pc * Hp + yy + p0) * Wp + xx + p0] * inv_scale));
        a = fmaf(d, p, a);
        c = fmaf(d, q, c);
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sh[0][0]; partials[2 * blockIdx.x + 1] = sh[1][0]; }
}

// The same for the diagonal kernels' scatter: the scattered term is the sum, in the order of the copies, of the
// `copies` copies of the padded gradient planes (copy_stride floats apart) the scatter stored into.
template <int COPIES>
__global__ void __launch_bounds__(kThreads)
diag_backward_finish_copies_kernel(const float *__restrict__ dy, const float *__restrict__ x, const float *__restrict__ P,
                                   const float *__restrict__ Q, const float *__restrict__ dxc, size_t copy_stride,
                                   const float *__restrict__ pos_w, const float *__restrict__ col_w, float *__restrict__ dx,
                                   float *__restrict__ partials, int planes, int H, int W, int Hp, int Wp, int p0)
{
    __shared__ float sh[2][kThreads];
    const float pw = *pos_w, cw = *col_w;
    const size_t n = (size_t)planes * H * W;
    float a = 0.f, c = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W);
        const size_t rest = i / W;
        const int yy = (int)(rest % H);
        const size_t pc = rest / H;
        const float p = P[i], q = Q[i], g = dy[i], d = g * x[i];
        const float *src = dxc + (pc * Hp + yy + p0) * Wp + xx + p0;
        float v[COPIES];
#pragma unroll
        for (int s = 0; s < COPIES; ++s) v[s] = __builtin_nontemporal_load(src + (size_t)s * copy_stride);
        float sc = 0.f;
#pragma unroll
        for (int s = 0; s < COPIES; ++s) sc += v[s];
        dx[i] = fmaf(g, fmaf(pw, p, cw * q), sc);
        a = fmaf(d, p, a);
        c = fmaf(d, q, c);
    }
    sh[0][threadIdx.x] = a;
    sh[1][threadIdx.x] = c;
    __syncthreads();
    for (int s = kThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) {
            sh[0][threadIdx.x] += sh[0][threadIdx.x + s];
            sh[1][threadIdx.x] += sh[1][threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) { partials[2 * blockIdx.x] = sh[0][0]; partials[2 * blockIdx.x + 1] = sh[1][0]; }
}

void launch_finish_copies(int k, int blocks, hipStream_t st, const float *dy, const float *x, const float *P, const float *Q,
                          const float *dxc, size_t copy_stride, const float *pos_w, const float *col_w, float *dx, float *partials,
                          int planes, int H, int W, int Hp, int Wp, int p0)
{
#define MFL_FINISH_CASE(KV)                                                                                                       \
    case KV:                                                                                                                      \
        hipLaunchKernelGGL(diag_backward_finish_copies_kernel<2 * KV>, dim3(blocks), dim3(kThreads), 0, st, dy, x, P, Q, dxc,     \
                           copy_stride, pos_w, col_w, dx, partials, planes, H, W, Hp, Wp, p0);                                    \
        break;
    switch (k) {
        MFL_FINISH_CASE(3) MFL_FINISH_CASE(5) MFL_FINISH_CASE(7) MFL_FINISH_CASE(9) MFL_FINISH_CASE(11)
    }
#undef MFL_FINISH_CASE
}

// dst[CE][CE] = src[C][C] zero-padded: constant row pitch for the fused epilogue's scalar loads
// dst[c2][c] = src[c][c2], zero-padded to CE2 x CE2 (CE2 even): the layout the fused epilogue reads
__global__ void pad_compat_kernel(const float *__restrict__ src, float *__restrict__ dst, int C, int CE2)
{
    for (int i = threadIdx.x; i < CE2 * CE2; i += blockDim.x) {
        const int c2 = i / CE2, c = i - c2 * CE2;
        dst[i] = (c < C && c2 < C) ? src[c * C + c2] : 0.f;
    }
}

// dst[c2][c] = src[c][c2] zero-padded to 32 x 32 (no matrix: the identity): the layout the stream
// kernels' fused epilogue loads its matrix operand from
__global__ void ring_compat_kernel(const float *__restrict__ src, float *__restrict__ dst, int C)
{
    for (int i = threadIdx.x; i < 32 * 32; i += blockDim.x) {
        const int c2 = i / 32, c = i - c2 * 32;
        dst[i] = (c < C && c2 < C) ? (src ? src[c * C + c2] : (c == c2 ? 1.f : 0.f)) : 0.f;
    }
}

size_t diag_padded_floats(const DiagGeom &g) { return (size_t)g.FS + (size_t)g.B * g.C * g.plane + 3 * (size_t)g.Wp + 64; }
// floats between the copies of the padded gradient planes the backward scatter stores into
size_t diag_copy_stride(const DiagGeom &g) { return (diag_padded_floats(g) + 63) / 64 * 64; }

DiagWorkspace diag_carve(int op, const DiagGeom &g)
{
    DiagWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    take(kTagBytes);                                  // plan tag
    w.xp = take(diag_padded_floats(g) * 4);
    if (op == MFL_OP_MEANFIELD_FORWARD) w.xp2 = take(diag_padded_floats(g) * 4);
    w.ip4 = take(((size_t)g.B * g.plane + 64) * sizeof(F4));
    w.table = take(diag_table_dwords(g) * 4);
    w.groups = take((size_t)g.k * g.k * 4);
    w.compat = take((size_t)(g.CH * g.NCS + 1) * (g.CH * g.NCS + 1) * 4);
    if (op == MFL_OP_LOCAL_BACKWARD) {
        w.xp2 = take(diag_copy_stride(g) * 2 * g.k * 4);  // 2k copies of the padded gradient planes (one per side and window row)
        w.psum = take((size_t)g.B * g.C * g.L * 4);
        w.qsum = take((size_t)g.B * g.C * g.L * 4);
        w.consts = take(8 * 4);
        w.partials = take((size_t)2 * kDiagBackwardBlocks * 4);
    }
    w.total = off;
    return w;
}

// The backward scatter stores each lane's sums into the copy of the gradient planes named by (side, window row): an
// element's k contributors of one line are k output pixels (C k k - 1) apart, and where that chain crosses the end of
// the pixel range it continues in the next tap.  With at least k C k k pixels it crosses at most once, so a store
// target has at most two writers, both lanes with pixels outside the image; those lanes add atomically (two
// floating-point addends onto zero commute).
bool diag_scatter_chain_ok(int C, int H, int W, int k) { return (long long)H * W >= (long long)k * C * k * k; }

bool diag_eligible(int op, int B, int C, int H, int W, int k, DiagGeom &g)
{
    if (op != MFL_OP_LOCAL_FORWARD && op != MFL_OP_MEANFIELD_FORWARD && op != MFL_OP_LOCAL_BACKWARD) return false;
    if (op == MFL_OP_LOCAL_BACKWARD && !diag_scatter_chain_ok(C, H, W, k)) return false;
    if (op == MFL_OP_MEANFIELD_FORWARD && C > 32) return false;
    return make_diag_geom(g, B, C, H, W, k, kDiagWeightLds, op != MFL_OP_LOCAL_BACKWARD) == 0;
}

// More than 33 labels: the diagonal kernels compute the window sums block by block of output channels (every
// block reads all input planes; one address table per block), for the mean-field loop whose label step runs as
// a separate kernel anyway.  All b-blocks run as regular workgroups (no tail workgroups).
struct BlockWorkspace {
    size_t xp, xp2, ip4, table, groups, msg, total;
    size_t table_stride;                      // bytes between the tables of consecutive blocks
    int nblk;
};

bool diag_block_eligible(int B, int C, int H, int W, int k, DiagGeom &g)
{
    if (C <= 33) return false;
    return make_diag_geom(g, B, C, H, W, k, kDiagWeightLds, false, diag_block_width(C)) == 0;
}

BlockWorkspace block_carve(int op, const DiagGeom &g)
{
    BlockWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    take(kTagBytes);                                  // plan tag
    w.nblk = (g.C + g.Cb - 1) / g.Cb;
    w.xp = take(diag_padded_floats(g) * 4);
    if (op == MFL_OP_MEANFIELD_FORWARD) w.xp2 = take(diag_padded_floats(g) * 4);
    w.ip4 = take(((size_t)g.B * g.plane + 64) * sizeof(F4));
    w.table_stride = align_up(diag_table_dwords(g) * 4, 256);
    w.table = take(w.table_stride * w.nblk);
    w.groups = take((size_t)g.k * g.k * 4);
    if (op == MFL_OP_MEANFIELD_FORWARD) w.msg = take((size_t)g.B * g.C * g.L * 4);
    w.total = off;
    return w;
}

// The local backward block by block of output channels (more than 33 labels): the forward's padded planes, image and
// per-block tables, plus what the diagonal backward adds (the 2k copies of the gradient planes, plain window sums, constants).
struct BlockBackwardWorkspace {
    BlockWorkspace f;
    size_t dxp, psum, qsum, consts, partials, total;
};

BlockBackwardWorkspace block_backward_carve(const DiagGeom &g)
{
    BlockBackwardWorkspace w{};
    w.f = block_carve(MFL_OP_LOCAL_FORWARD, g);
    size_t off = w.f.total;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.dxp = take(diag_copy_stride(g) * 2 * g.k * 4);
    w.psum = take((size_t)g.B * g.C * g.L * 4);
    w.qsum = take((size_t)g.B * g.C * g.L * 4);
    w.consts = take(8 * 4);
    w.partials = take((size_t)2 * kDiagBackwardBlocks * 4);
    w.total = off;
    return w;
}

// The geometry of block `blk` of a block geometry.
DiagGeom block_geom(const DiagGeom &g, int blk)
{
    DiagGeom b = g;
    b.c0 = blk * g.Cb;
    b.Cv = g.C - b.c0 < g.Cb ? g.C - b.c0 : g.Cb;
    return b;
}

int dispatch_diag(const DiagArgs &a, int mode, hipStream_t st)
{
#define MFL_DIAG_CASE(CHV, KSV) \
    if (a.g.CH == CHV && a.g.k == KSV) return launch_diag_ch##CHV##_k##KSV(a, mode, st);
#define MFL_DIAG_CASES(KSV) MFL_DIAG_CASE(8, KSV) MFL_DIAG_CASE(10, KSV) MFL_DIAG_CASE(11, KSV)
    MFL_DIAG_CASES(3) MFL_DIAG_CASES(5) MFL_DIAG_CASES(7) MFL_DIAG_CASE(5, 9) MFL_DIAG_CASE(5, 11)
#undef MFL_DIAG_CASES
#undef MFL_DIAG_CASE
    return MFL_ERR_INVALID_ARGUMENT;
}

// Pads x (planes) and img, builds the address table and the weight-group list.  x_floats: the padded
// buffer (front slack included) that receives x.
int diag_prepare(int op, void *workspace, const DiagGeom &g, const float *x, const float *img, float *x_floats, F4 *ip4,
                 uint32_t *table, uint32_t *groups_dev, hipStream_t st)
{
    const size_t n = (size_t)g.B * g.C * g.plane;
    MFL_CHECK_HIP(hipMemsetAsync(x_floats, 0, (size_t)g.FS * 4, st));
    hipLaunchKernelGGL(pad_planes_kernel, dim3(pad_rows_grid((size_t)g.B * g.C * g.Hp)), dim3(kThreads), 0, st, x, x_floats + g.FS, g.B * g.C, g.H,
                       g.W, g.Hp, g.Wp, g.p0);
    MFL_CHECK_HIP(hipMemsetAsync(x_floats + g.FS + n, 0, (3 * (size_t)g.Wp + 64) * 4, st));
    const size_t ni = (size_t)g.B * g.plane;
    hipLaunchKernelGGL(pad_image4_kernel, dim3(grid_for(ni)), dim3(kThreads), 0, st, img, ip4, g.B, g.H, g.W, g.Hp,
                       g.Wp, g.p0);
    MFL_CHECK_HIP(hipMemsetAsync(ip4 + ni, 0, 64 * sizeof(F4), st));
    return launch_diag_table(g, table, groups_dev, (uint32_t *)workspace, make_tag(2, op, g.B, g.C, g.H, g.W, g.k), st);
}

void diag_lds(const DiagGeom &g, DiagArgs &a)
{
    const size_t wbytes = 2 * (size_t)g.TB * g.NBR * g.RUNP * 4;
#ifdef MFL_EXP_NGW
    const size_t lds = wbytes > 24 * 1024 ? wbytes : 24 * 1024;   // timing experiment: several workgroups per CU
#else
    const size_t lds = wbytes > 96 * 1024 ? wbytes : 96 * 1024;
#endif
    // the epilogue passes the workgroup's b rows through LDS in equal chunks (fewest that fit)
    const int max_rows = (int)(lds / ((size_t)g.CH * g.NCS * g.RUNP * 4));
    const int chunks = (g.NBR + max_rows - 1) / max_rows;
    const int rows = (g.NBR + chunks - 1) / chunks;
    a.lds_bytes = lds;
    a.epi_rows = rows;
}

// Pads x (planes) into xp and img into ip4 and builds the address tables of all channel blocks (unless the
// workspace holds them already).
int block_prepare(int op, void *workspace, const DiagGeom &bg, const BlockWorkspace &ws, const float *x, const float *img,
                  float *xp, hipStream_t st)
{
    char *base = (char *)workspace;
    F4 *ip4 = (F4 *)(base + ws.ip4);
    const size_t n = (size_t)bg.B * bg.C * bg.plane;
    MFL_CHECK_HIP(hipMemsetAsync(xp, 0, (size_t)bg.FS * 4, st));
    hipLaunchKernelGGL(pad_planes_kernel, dim3(pad_rows_grid((size_t)bg.B * bg.C * bg.Hp)), dim3(kThreads), 0, st, x, xp + bg.FS,
                       bg.B * bg.C, bg.H, bg.W, bg.Hp, bg.Wp, bg.p0);
    MFL_CHECK_HIP(hipMemsetAsync(xp + bg.FS + n, 0, (3 * (size_t)bg.Wp + 64) * 4, st));
    const size_t ni = (size_t)bg.B * bg.plane;
    hipLaunchKernelGGL(pad_image4_kernel, dim3(grid_for(ni)), dim3(kThreads), 0, st, img, ip4, bg.B, bg.H, bg.W, bg.Hp, bg.Wp, bg.p0);
    MFL_CHECK_HIP(hipMemsetAsync(ip4 + ni, 0, 64 * sizeof(F4), st));
    uint32_t *tag = (uint32_t *)workspace;
    const PlanTag want = make_tag(4, op, bg.B, bg.C, bg.H, bg.W, bg.k);
    hipLaunchKernelGGL(diag_groups_kernel, dim3(1), dim3(64), 0, st, bg, (uint32_t *)(base + ws.groups), tag, want);
    const size_t nrec = (size_t)bg.NBB * bg.NG * bg.K * bg.NCS;
    const unsigned blocks = (unsigned)((nrec + 255) / 256 < 65535u * 8 ? (nrec + 255) / 256 : 65535u * 8);
    for (int blk = 0; blk < ws.nblk; ++blk)
        hipLaunchKernelGGL(diag_table_kernel, dim3(blocks), dim3(256), 0, st, block_geom(bg, blk),
                           (uint32_t *)(base + ws.table + blk * ws.table_stride), tag, want);
    hipLaunchKernelGGL(stamp_tag_kernel, dim3(1), dim3(64), 0, st, tag, want);
    return (int)hipGetLastError();
}

// y = x * window sums of the padded planes xp, one launch per channel block.
int block_window_products(void *workspace, const DiagGeom &bg, const BlockWorkspace &ws, const float *xp, const float *pos_w,
                          const float *col_w, float *y, hipStream_t st, int mode = 1)
{
    char *base = (char *)workspace;
    DiagArgs da{};
    diag_lds(bg, da);
    da.xp = xp;
    da.ip4 = (const F4 *)(base + ws.ip4);
    da.groups = (const uint32_t *)(base + ws.groups);
    da.pos_w = pos_w;
    da.col_w = col_w;
    da.y = y;
    for (int blk = 0; blk < ws.nblk; ++blk) {
        da.g = block_geom(bg, blk);
        da.table = (const uint32_t *)(base + ws.table + blk * ws.table_stride);
        const int rc = dispatch_diag(da, mode, st);
        if (rc) return rc;
    }
    return MFL_OK;
}


// ---- ring gather path (mfl_ring.hip): the forward operators for images at least 64 + k wide, odd k <= 11 ----
__global__ void ring_table_kernel(RingGeom r, uint32_t *__restrict__ table, const uint32_t *__restrict__ tag, PlanTag want)
{
    if (tag_matches(tag, want)) return;               // the workspace holds this plan already
    const DiagGeom &g = r.d;
    const size_t nrec = (size_t)g.NBB * g.NG * r.recs_per_wave * g.NCS;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nrec; i += (size_t)gridDim.x * blockDim.x) {
        const int ci = (int)(i % g.NCS);
        size_t rest = i / g.NCS;
        const int pos = (int)(rest % r.recs_per_wave);
        rest /= r.recs_per_wave;
        const int gamma = (int)(rest % g.NG);
        const int beta = (int)(rest / g.NG);
        ring_fill_record(r, beta, gamma, pos, ci, table + i * g.RECW);
    }
}

struct RingWorkspace {
    size_t xp, xp2, ip4, table, wtab, compat, total;
};

RingWorkspace ring_carve(int op, const RingGeom &r)
{
    const DiagGeom &g = r.d;
    RingWorkspace w{};
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    take(kTagBytes);                                  // plan tag
    w.xp = take(diag_padded_floats(g) * 4);
    if (op == MFL_OP_MEANFIELD_FORWARD) w.xp2 = take(diag_padded_floats(g) * 4);
    w.ip4 = take(((size_t)g.B * g.plane + 64) * sizeof(F4));
    w.table = take(ring_table_dwords(r) * 4 + 256);
    w.wtab = take(ring_weight_floats(r) * 4 + 256);
    w.compat = take(32 * 32 * 4);
    w.total = off;
    return w;
}

bool ring_eligible(int op, int B, int C, int H, int W, int k, RingGeom &r)
{
    if (op != MFL_OP_LOCAL_FORWARD && op != MFL_OP_MEANFIELD_FORWARD) return false;
#ifdef MFL_NO_RING
    return false;
#endif
    if (make_ring_geom(r, B, C, H, W, k) != 0) return false;
#ifdef MFL_EXPERIMENTS
    if (const char *e = getenv("MFL_RING_NGW")) {        // timing experiments: another workgroup width
        DiagGeom &g = r.d;
        g.NGW = atoi(e);
        g.NGB = (g.NG + g.NGW - 1) / g.NGW;
        g.RUN = g.NGW * g.k;
        g.RUNP = (g.RUN & 1) ? g.RUN + 2 : g.RUN + 1;
    }
#endif
    return true;
}

int dispatch_ring(const RingArgs &a, int mode, hipStream_t st)
{
#define MFL_RING_CASE(CHV, KSV) \
    if (a.r.d.CH == CHV && a.r.d.k == KSV) return launch_ring_ch##CHV##_k##KSV(a, mode, st);
    MFL_RING_CASE(8, 9) MFL_RING_CASE(7, 11) MFL_RING_CASE(10, 9) MFL_RING_CASE(10, 11)
#undef MFL_RING_CASE
    return MFL_ERR_INVALID_ARGUMENT;
}

// Pads x (planes) and img and builds the record table.
int ring_prepare(int op, void *workspace, const RingGeom &r, const float *x, const float *img, const float *pos_w,
                 const float *col_w, float *x_floats, F4 *ip4, uint32_t *table, float *wtab, hipStream_t st)
{
    const DiagGeom &g = r.d;
    const size_t n = (size_t)g.B * g.C * g.plane;
    MFL_CHECK_HIP(hipMemsetAsync(x_floats, 0, (size_t)g.FS * 4, st));
    hipLaunchKernelGGL(pad_planes_kernel, dim3(pad_rows_grid((size_t)g.B * g.C * g.Hp)), dim3(kThreads), 0, st, x, x_floats + g.FS,
                       g.B * g.C, g.H, g.W, g.Hp, g.Wp, g.p0);
    MFL_CHECK_HIP(hipMemsetAsync(x_floats + g.FS + n, 0, (3 * (size_t)g.Wp + 64) * 4, st));
    const size_t ni = (size_t)g.B * g.plane;
    hipLaunchKernelGGL(pad_image4_kernel, dim3(grid_for(ni)), dim3(kThreads), 0, st, img, ip4, g.B, g.H, g.W, g.Hp,
                       g.Wp, g.p0);
    MFL_CHECK_HIP(hipMemsetAsync(ip4 + ni, 0, 64 * sizeof(F4), st));
    const size_t nrec = (size_t)g.NBB * g.NG * r.recs_per_wave * g.NCS;
    const unsigned blocks = (unsigned)((nrec + 255) / 256 < 65535u * 8 ? (nrec + 255) / 256 : 65535u * 8);
    const PlanTag want = make_tag(1, op, g.B, g.C, g.H, g.W, g.k);
    hipLaunchKernelGGL(ring_table_kernel, dim3(blocks), dim3(256), 0, st, r, table, (const uint32_t *)workspace, want);
    hipLaunchKernelGGL(stamp_tag_kernel, dim3(1), dim3(64), 0, st, (uint32_t *)workspace, want);
    RingArgs wa{};
    wa.r = r;
    wa.ip4 = ip4;
    wa.pos_w = pos_w;
    wa.col_w = col_w;
    wa.wtab_out = wtab;
    return dispatch_ring(wa, 3, st);
}

}  // namespace

// ---- built-in scratch of mfl_meanfield_forward_noalloc --------------------------------------------
// Part of the library's device image: every GPU the library runs on holds its own copy, at its own
// address, until the process ends.  `scratch` below is a host-side stand-in for the symbol, not a device
// address: host code only ever uses the pointer hipGetSymbolAddress returned for the current device.
static __device__ __align__(256) unsigned char scratch[MFL_NOALLOC_SCRATCH_BYTES];

namespace {

constexpr int kMaxDevices = 64;
std::atomic<void *> g_scratch_ptr[kMaxDevices];

// Address of the current device's copy of `scratch` (looked up once per device, with that device current).
int scratch_pointer(void **out)
{
    int dev = -1;
    MFL_CHECK_HIP(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return MFL_ERR_INVALID_ARGUMENT;
    void *p = g_scratch_ptr[dev].load(std::memory_order_acquire);
    if (!p) {
        MFL_CHECK_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (!p) return MFL_ERR_INVALID_ARGUMENT;
        g_scratch_ptr[dev].store(p, std::memory_order_release);   // (racing first calls store the same address)
    }
    *out = p;
    return MFL_OK;
}

}  // namespace

namespace mfl {
// The diagonal local backward in two phases sharing one workspace (the loop backward puts its own kernel between
// them): sums = padding, tables and the two plain window sums P, Q (they do not depend on dy); apply = scatter of
// dy * x, dx and the parameter gradients.
bool local_backward_split_supported(int B, int C, int H, int W, int k);
int local_backward_sums(const float *x, const float *img, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                        mfl_stream_t stream, const float **P, const float **Q);
int local_backward_apply(const float *x, const float *img, const float *pos_w, const float *col_w, const float *dy, float *dx,
                         float *dparams, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                         mfl_stream_t stream);
size_t loop_backward_workspace_bytes(int B, int C, int H, int W, int k, int iters);   // mfl_loop_backward.hip
bool label_step_supported(int C);                                                    // mfl_label_step.hip
bool label_step_fits(int B, int C, int H, int W, int q_pad_k);
int launch_label_step(const float *msg, const float *aw, const float *ab, const float *compat, const float *unary, float *logq,
                      float *q, int q_pad_k, int B, int C, int H, int W, hipStream_t st);
}

// ------------------------------------------------------------------ C ABI

extern "C" {

#ifdef MFL_STAMPS
void mfl_debug_stamps(void *p) { g_stamps = (unsigned long long *)p; }
#endif

int mfl_version(void) { return MFL_VERSION; }

const char *mfl_error_string(int code)
{
    switch (code) {
    case MFL_OK: return "success";
    case MFL_ERR_INVALID_ARGUMENT: return "mfl: invalid argument";
    case MFL_ERR_WORKSPACE: return "mfl: workspace missing or too small";
    case MFL_ERR_TOO_LARGE: return "mfl: problem too large for 32-bit kernel indexing";
    case MFL_ERR_SCRATCH_LIMIT: return "mfl: problem needs more scratch than the library's built-in buffer (MFL_NOALLOC_SCRATCH_BYTES)";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "mfl: unknown error";
    }
}

size_t mfl_workspace_bytes(int op, int B, int C, int H, int W, int k, int iters)
{
    if (op == MFL_OP_MEANFIELD_BACKWARD) return mfl::loop_backward_workspace_bytes(B, C, H, W, k, iters);
    if (op != MFL_OP_LOCAL_FORWARD && op != MFL_OP_LOCAL_BACKWARD && op != MFL_OP_MEANFIELD_FORWARD) return 0;
    RingGeom rg;
    if (ring_eligible(op, B, C, H, W, k, rg) && !(op == MFL_OP_MEANFIELD_FORWARD && C > 32)) {
        // (the local forward takes the ring kernels with an image and the generic ones with a cd tensor)
        size_t need = ring_carve(op, rg).total;
        if (op == MFL_OP_LOCAL_FORWARD) {
            Geom g0;
            if (make_geom(g0, B, C, H, W, k, 1) == 0) {
                const size_t generic = carve(op, g0, pick_cc(C)).total;
                if (generic > need) need = generic;
            }
        }
        return need;
    }
    DiagGeom dg;
    const bool diag = diag_eligible(op, B, C, H, W, k, dg);
    if (diag && op != MFL_OP_LOCAL_BACKWARD) return diag_carve(op, dg).total;
    Geom g;
    if (make_geom(g, B, C, H, W, k, 1)) return 0;
    size_t generic = carve(op, g, pick_cc(C)).total;
    if (diag_block_eligible(B, C, H, W, k, dg) && (op != MFL_OP_LOCAL_BACKWARD || diag_scatter_chain_ok(C, H, W, k))) {
        // (more than 33 labels: the diagonal kernels block by block with an image, and in the loop with a label
        // matrix; the generic ones otherwise)
        const size_t blocks = op == MFL_OP_LOCAL_BACKWARD ? block_backward_carve(dg).total : block_carve(op, dg).total;
        if (blocks > generic) generic = blocks;
    }
    // the backward takes the diagonal kernels with an image and the generic ones with a cd tensor
    if (diag) { const size_t d = diag_carve(op, dg).total; return d > generic ? d : generic; }
    return generic;
}

int mfl_color_distance(const float *img, float *cd, int B, int H, int W, int k, mfl_stream_t stream)
{
    int rc = check_common(img, cd, B, 1, H, W, k);
    if (rc) return rc;
    const size_t n = (size_t)B * H * W * k * k;
    hipLaunchKernelGGL(color_distance_kernel, dim3(grid_for(n)), dim3(kThreads), 0, (hipStream_t)stream, img, cd, B, H,
                       W, k);
    return (int)hipGetLastError();
}

int mfl_local_forward(const float *x, const float *img, const float *cd, const float *pos_w, const float *col_w,
                      float *y, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                      mfl_stream_t stream)
{
    int rc = check_common(x, y, B, C, H, W, k);
    if (rc) return rc;
    if (!pos_w || !col_w || (img == nullptr) == (cd == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    RingGeom rg;
    if (img && ring_eligible(MFL_OP_LOCAL_FORWARD, B, C, H, W, k, rg)) {
        const RingWorkspace ws = ring_carve(MFL_OP_LOCAL_FORWARD, rg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        rc = ring_prepare(MFL_OP_LOCAL_FORWARD, workspace, rg, x, img, pos_w, col_w, (float *)(base + ws.xp), (F4 *)(base + ws.ip4), (uint32_t *)(base + ws.table),
                          (float *)(base + ws.wtab), st);
        if (rc) return rc;
        RingArgs ra{};
        ra.r = rg;
        ra.xp = (const float *)(base + ws.xp);
        ra.ip4 = (const F4 *)(base + ws.ip4);
        ra.table = (const uint32_t *)(base + ws.table);
        ra.wtab = (const float *)(base + ws.wtab);
        ra.pos_w = pos_w;
        ra.col_w = col_w;
        ra.y = y;
        return dispatch_ring(ra, 1, st);
    }
    DiagGeom dg;
    if (img && diag_eligible(MFL_OP_LOCAL_FORWARD, B, C, H, W, k, dg)) {
        const DiagWorkspace ws = diag_carve(MFL_OP_LOCAL_FORWARD, dg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        rc = diag_prepare(MFL_OP_LOCAL_FORWARD, workspace, dg, x, img, (float *)(base + ws.xp), (F4 *)(base + ws.ip4), (uint32_t *)(base + ws.table),
                          (uint32_t *)(base + ws.groups), st);
        if (rc) return rc;
        DiagArgs da{};
        da.g = dg;
        diag_lds(dg, da);
        MFL_SET_DBG(da);
        da.xp = (const float *)(base + ws.xp);
        da.ip4 = (const F4 *)(base + ws.ip4);
        da.table = (const uint32_t *)(base + ws.table);
        da.groups = (const uint32_t *)(base + ws.groups);
        da.pos_w = pos_w;
        da.col_w = col_w;
        da.y = y;
        return dispatch_diag(da, 1, st);
    }
    if (img && diag_block_eligible(B, C, H, W, k, dg)) {
        // more than 33 labels: the diagonal kernels, one launch per block of output channels
        const BlockWorkspace ws = block_carve(MFL_OP_LOCAL_FORWARD, dg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        hipStream_t st = (hipStream_t)stream;
        float *xp = (float *)((char *)workspace + ws.xp);
        rc = block_prepare(MFL_OP_LOCAL_FORWARD, workspace, dg, ws, x, img, xp, st);
        if (rc) return rc;
        return block_window_products(workspace, dg, ws, xp, pos_w, col_w, y, st);
    }
    Geom g;
    int cc;
    bool tab_lds;
    rc = setup(g, cc, tab_lds, B, C, H, W, k, cd != nullptr, 4, 0);
    if (rc) return rc;
    const Workspace ws = carve(MFL_OP_LOCAL_FORWARD, g, cc);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    float *xp = (float *)(base + ws.xp);
    F4 *ip4 = (F4 *)(base + ws.ip4);
    int *tab = (int *)(base + ws.tab);
    hipStream_t st = (hipStream_t)stream;
    rc = launch_prepare(workspace, g, x, img, xp, ip4, tab, st);
    if (rc) return rc;
    const GatherArgs ga{g, cd != nullptr, tab_lds, xp, ip4, cd, tab, pos_w, col_w};
    return dispatch_local_forward(cc, ga, y, st);
}

int mfl_local_backward(const float *x, const float *img, const float *cd, const float *pos_w, const float *col_w,
                       const float *dy, float *dx, float *dparams, float *dcd, int B, int C, int H, int W, int k,
                       void *workspace, size_t workspace_bytes, mfl_stream_t stream)
{
    int rc = check_common(x, dy, B, C, H, W, k);
    if (rc) return rc;
    if (!dx || !dparams || !pos_w || !col_w || (img == nullptr) == (cd == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (dcd && !cd) return MFL_ERR_INVALID_ARGUMENT;
    if (img && mfl::local_backward_split_supported(B, C, H, W, k)) {
        // dx = dy * S + scatter, S = pos_w * P + col_w * Q: two gathers with the weights (1, 0) and
        // (0, 1) give P and Q (and with them both parameter gradients), one scatter launch the rest
        const float *P, *Q;
        rc = mfl::local_backward_sums(x, img, B, C, H, W, k, workspace, workspace_bytes, stream, &P, &Q);
        if (rc) return rc;
        return mfl::local_backward_apply(x, img, pos_w, col_w, dy, dx, dparams, B, C, H, W, k, workspace, workspace_bytes, stream);
    }
    Geom g;
    int cc;
    bool tab_lds;
    rc = setup(g, cc, tab_lds, B, C, H, W, k, cd != nullptr, 2, 0);
    if (rc) return rc;
    const Workspace ws = carve(MFL_OP_LOCAL_BACKWARD, g, cc);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    float *xp = (float *)(base + ws.xp);
    F4 *ip4 = (F4 *)(base + ws.ip4);
    int *tab = (int *)(base + ws.tab);
    float *dxd = (float *)(base + ws.dxd);
    float *dxp = (float *)(base + ws.dxp);
    float *partials = (float *)(base + ws.partials);
    hipStream_t st = (hipStream_t)stream;
    rc = launch_prepare(workspace, g, x, img, xp, ip4, tab, st);
    if (rc) return rc;
    if (img && taps_kernel_applies(g)) {
        // large windows on small images: the taps of a pixel split over threads, plain window sums P, Q and an
        // order-free fixed-point scatter, then the finishing pass the diagonal backward uses
        float *consts = (float *)(base + ws.consts), *P = (float *)(base + ws.psum), *Q = (float *)(base + ws.qsum);
        unsigned long long *dxp64 = (unsigned long long *)(base + ws.dxp64);
        float *partials2 = (float *)(base + ws.partials2);
        const size_t n = (size_t)B * C * g.L;
        hipLaunchKernelGGL(diag_backward_consts_kernel, dim3(1), dim3(64), 0, st, consts);
        hipLaunchKernelGGL(diag_backward_absmax_kernel, dim3(grid_for(n) < 1024 ? grid_for(n) : 1024), dim3(kThreads), 0, st, dy, x, n, img,
                           (size_t)B * 3 * g.L, consts);
        hipLaunchKernelGGL(diag_backward_scale_kernel, dim3(1), dim3(64), 0, st, pos_w, col_w, k, consts);
        MFL_CHECK_HIP(hipMemsetAsync(dxp64, 0, padded_floats(g) * 8, st));
        const GatherArgs ga{g, false, false, xp, ip4, nullptr, tab, pos_w, col_w};
        const TapsBackwardArgs ta{dy, consts + 4, P, Q, dxp64};
        rc = dispatch_local_backward_taps(cc, ga, ta, st);
        if (rc) return rc;
        const int blocks = grid_for(n) < kDiagBackwardBlocks ? grid_for(n) : kDiagBackwardBlocks;
        hipLaunchKernelGGL(diag_backward_finish_kernel, dim3(blocks), dim3(kThreads), 0, st, dy, x, P, Q, (const long long *)dxp64, consts,
                           pos_w, col_w, dx, partials2, B * C, H, W, g.Hp, g.Wp, g.p0);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kThreads), 0, st, partials2, blocks, dparams);
        return (int)hipGetLastError();
    }
    MFL_CHECK_HIP(hipMemsetAsync(dxp, 0, padded_floats(g) * 4, st));
    const int nchunks = (g.C + cc - 1) / cc;
    if (dcd && nchunks > 1) MFL_CHECK_HIP(hipMemsetAsync(dcd, 0, (size_t)B * g.L * g.K * 4, st));
    const GatherArgs ga{g, cd != nullptr, tab_lds, xp, ip4, cd, tab, pos_w, col_w};
    const BackwardArgs ba{dy, dxd, dxp, partials, dcd};
    rc = dispatch_local_backward(cc, ga, ba, st);
    if (rc) return rc;
    const size_t n = (size_t)B * C * g.L;
    hipLaunchKernelGGL(backward_finish_kernel, dim3(grid_for(n)), dim3(kThreads), 0, st, dxd, dxp, dx, B * C, H, W,
                       g.Hp, g.Wp, g.p0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kThreads), 0, st, partials, backward_blocks(g, cc),
                       dparams);
    return (int)hipGetLastError();
}

int mfl_combine_forward(const float *unary, const float *smooth, int mixtures, float *logq, float *q, int B, int C,
                        int H, int W, mfl_stream_t stream)
{
    int rc = check_common(unary, smooth, B, C, H, W, 1);
    if (rc) return rc;
    if (!logq || mixtures <= 0) return MFL_ERR_INVALID_ARGUMENT;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, unary, smooth, mixtures, logq, q, 0, B, C, H, W);
    return (int)hipGetLastError();
}

// The mean-field loop.  logq_all (optional): receives logQ_{t+1} of every iteration t at
// logq_all + t * B*C*H*W (the loop backward re-reads them); logq (optional then) the last one.
static int meanfield_forward_impl(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                  const float *affine_w, const float *affine_b, const float *compat, float *logq,
                                  float *logq_all, int B, int C, int H, int W, int k, int iters, void *workspace,
                                  size_t workspace_bytes, mfl_stream_t stream)
{
    int rc = check_common(unary, img, B, C, H, W, k);
    if (rc) return rc;
    if ((!logq && !logq_all) || !pos_w || !col_w || iters <= 0 || (affine_w == nullptr) != (affine_b == nullptr))
        return MFL_ERR_INVALID_ARGUMENT;
    const size_t n_all = (size_t)B * C * H * W;
    auto out_of = [&](int it) -> float * {
        if (logq_all) return logq_all + (size_t)it * n_all;
        return it == iters - 1 ? logq : nullptr;
    };
    RingGeom rg;
    if (C <= 32 && ring_eligible(MFL_OP_MEANFIELD_FORWARD, B, C, H, W, k, rg)) {
        const DiagGeom &g = rg.d;
        const RingWorkspace ws = ring_carve(MFL_OP_MEANFIELD_FORWARD, rg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        float *qa = (float *)(base + ws.xp), *qb = (float *)(base + ws.xp2);
        rc = ring_prepare(MFL_OP_MEANFIELD_FORWARD, workspace, rg, unary, img, pos_w, col_w, qa, (F4 *)(base + ws.ip4), (uint32_t *)(base + ws.table),
                          (float *)(base + ws.wtab), st);
        if (rc) return rc;
        MFL_CHECK_HIP(hipMemsetAsync(qb, 0, diag_padded_floats(g) * 4, st));   // borders of the other buffer
        RingArgs ra{};
        ra.r = rg;
        ra.ip4 = (const F4 *)(base + ws.ip4);
        ra.table = (const uint32_t *)(base + ws.table);
        ra.wtab = (const float *)(base + ws.wtab);
        ra.pos_w = pos_w;
        ra.col_w = col_w;
        ra.unary = unary;
        ra.aw = affine_w;
        ra.ab = affine_b;
        {
            float *cpad = (float *)(base + ws.compat);
            hipLaunchKernelGGL(ring_compat_kernel, dim3(1), dim3(kThreads), 0, st, compat, cpad, C);
            ra.compat = cpad;
            ra.compat_plain = compat;
        }
#ifdef MFL_STAMPS
        ra.y = (float *)g_stamps;      // diagnostic builds: cycle stamps of the first workgroups
#endif
        for (int it = 0; it < iters; ++it) {
            ra.xp = qa;
            ra.qp_out = qb;
            ra.logq = out_of(it);
            rc = dispatch_ring(ra, 0, st);
            if (rc) return rc;
            float *t = qa; qa = qb; qb = t;
        }
        return MFL_OK;
    }
    DiagGeom dg;
    if (diag_eligible(MFL_OP_MEANFIELD_FORWARD, B, C, H, W, k, dg)) {
        const DiagWorkspace ws = diag_carve(MFL_OP_MEANFIELD_FORWARD, dg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        float *qa = (float *)(base + ws.xp), *qb = (float *)(base + ws.xp2);
        rc = diag_prepare(MFL_OP_MEANFIELD_FORWARD, workspace, dg, unary, img, qa, (F4 *)(base + ws.ip4), (uint32_t *)(base + ws.table),
                          (uint32_t *)(base + ws.groups), st);
        if (rc) return rc;
        MFL_CHECK_HIP(hipMemsetAsync(qb, 0, diag_padded_floats(dg) * 4, st));   // borders of the other buffer
        DiagArgs da{};
        da.g = dg;
        diag_lds(dg, da);
        MFL_SET_DBG(da);
        da.ip4 = (const F4 *)(base + ws.ip4);
        da.table = (const uint32_t *)(base + ws.table);
        da.groups = (const uint32_t *)(base + ws.groups);
        da.pos_w = pos_w;
        da.col_w = col_w;
        da.unary = unary;
        da.aw = affine_w;
        da.ab = affine_b;
        if (compat) {
            float *cpad = (float *)(base + ws.compat);
            hipLaunchKernelGGL(pad_compat_kernel, dim3(1), dim3(kThreads), 0, st, compat, cpad, C, (dg.CH * dg.NCS + 1) & ~1);
            da.compat = cpad;
            da.compat_plain = compat;
        }
        for (int it = 0; it < iters; ++it) {
            da.xp = qa;
            da.qp_out = qb;
            da.logq = out_of(it);
            rc = dispatch_diag(da, 0, st);
            if (rc) return rc;
            float *t = qa; qa = qb; qb = t;
        }
        return MFL_OK;
    }
    DiagGeom bg;
    if (C > 32 && compat && mfl::label_step_supported(C) && mfl::label_step_fits(B, C, H, W, k) &&
        diag_block_eligible(B, C, H, W, k, bg)) {
        // 33 .. 160 labels: window sums on the diagonal kernels, one launch per block of output channels, then the
        // label step (affine, label product on the matrix cores, normalise) as one kernel
        const BlockWorkspace ws = block_carve(MFL_OP_MEANFIELD_FORWARD, bg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        float *qa = (float *)(base + ws.xp), *qb = (float *)(base + ws.xp2);
        float *msg = (float *)(base + ws.msg);
        hipStream_t st = (hipStream_t)stream;
        rc = block_prepare(MFL_OP_MEANFIELD_FORWARD, workspace, bg, ws, unary, img, qa, st);      // Q_0 = unary, padded
        if (rc) return rc;
        MFL_CHECK_HIP(hipMemsetAsync(qb, 0, diag_padded_floats(bg) * 4, st));   // borders of the other buffer
        for (int it = 0; it < iters; ++it) {
            rc = block_window_products(workspace, bg, ws, qa, pos_w, col_w, msg, st);              // msg = Q_t * window sums
            if (rc) return rc;
            rc = mfl::launch_label_step(msg, affine_w, affine_b, compat, unary, out_of(it), qb + bg.FS, k, B, C, H, W, st);
            if (rc) return rc;
            float *t = qa; qa = qb; qb = t;
        }
        return MFL_OK;
    }
    // (large windows on small images: the tap-split forward kernel, then the unfused label step)
    bool fused = C <= 32;
    {
        Geom g1;
        if (fused && make_geom(g1, B, C, H, W, k, 1) == 0 && taps_kernel_applies(g1)) fused = false;
    }
    // the label transform of the unfused path stages 64 pixels x C messages in LDS (160 KB per workgroup)
    if (!fused && (affine_w || compat) && (size_t)C * 64 * 4 > 160 * 1024) return MFL_ERR_INVALID_ARGUMENT;
    Geom g;
    int cc;
    bool tab_lds;
    rc = setup(g, cc, tab_lds, B, C, H, W, k, false, 4, fused ? (size_t)C * C * 4 : 0);
    if (rc) return rc;
    const Workspace ws = carve(MFL_OP_MEANFIELD_FORWARD, g, cc);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    float *qa = (float *)(base + ws.xp), *qb = (float *)(base + ws.xp2);
    F4 *ip4 = (F4 *)(base + ws.ip4);
    int *tab = (int *)(base + ws.tab);
    float *msg = (float *)(base + ws.msg);
    hipStream_t st = (hipStream_t)stream;
    rc = launch_prepare(workspace, g, unary, img, qa, ip4, tab, st);          // Q_0 = unary, padded
    if (rc) return rc;
    MFL_CHECK_HIP(hipMemsetAsync(qb, 0, padded_floats(g) * 4, st)); // borders of the other buffer
    const size_t npix = (size_t)B * g.L;
    for (int it = 0; it < iters; ++it) {
        float *lq = out_of(it);
        if (fused) {
            const GatherArgs ga{g, false, tab_lds, qa, ip4, nullptr, tab, pos_w, col_w};
            const StepArgs sa{unary, affine_w, affine_b, compat, lq, qb};
            rc = dispatch_meanfield_step(cc, ga, sa, st);
            if (rc) return rc;
        } else {
            const GatherArgs ga{g, false, tab_lds, qa, ip4, nullptr, tab, pos_w, col_w};
            rc = dispatch_local_forward(cc, ga, msg, st);
            if (rc) return rc;
            if (compat && mfl::label_step_supported(C) && mfl::label_step_fits(B, C, H, W, k)) {
                // label product on the matrix cores, fused with the affine and the normalise step
                rc = mfl::launch_label_step(msg, affine_w, affine_b, compat, unary, lq, qb, k, B, C, H, W, st);
                if (rc) return rc;
                float *t = qa; qa = qb; qb = t;
                continue;
            }
            if (affine_w || compat) {
                // more than the default 64 KB of dynamic LDS past 256 labels (per device: set before every launch)
                MFL_CHECK_HIP(hipFuncSetAttribute((const void *)label_transform_kernel,
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
                hipLaunchKernelGGL(label_transform_kernel, dim3((unsigned)((npix + 63) / 64)), dim3(kThreads),
                                   (size_t)C * 64 * 4, st, msg, affine_w, affine_b, compat, msg, B, C, H, W);
            }
            hipLaunchKernelGGL(combine_kernel, dim3((unsigned)((npix + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                               st, unary, msg, 1, lq, qb, k, B, C, H, W);
            rc = (int)hipGetLastError();
            if (rc) return rc;
        }
        float *t = qa; qa = qb; qb = t;
    }
    return MFL_OK;
}

int mfl_meanfield_forward(const float *unary, const float *img, const float *pos_w, const float *col_w,
                          const float *affine_w, const float *affine_b, const float *compat, float *logq, int B,
                          int C, int H, int W, int k, int iters, void *workspace, size_t workspace_bytes,
                          mfl_stream_t stream)
{
    if (!logq) return MFL_ERR_INVALID_ARGUMENT;
    return meanfield_forward_impl(unary, img, pos_w, col_w, affine_w, affine_b, compat, logq, nullptr, B, C, H, W, k, iters,
                                  workspace, workspace_bytes, stream);
}

int mfl_meanfield_forward_iterates(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                   const float *affine_w, const float *affine_b, const float *compat, float *logq_iterates,
                                   int B, int C, int H, int W, int k, int iters, void *workspace, size_t workspace_bytes,
                                   mfl_stream_t stream)
{
    if (!logq_iterates) return MFL_ERR_INVALID_ARGUMENT;
    return meanfield_forward_impl(unary, img, pos_w, col_w, affine_w, affine_b, compat, nullptr, logq_iterates, B, C, H, W, k,
                                  iters, workspace, workspace_bytes, stream);
}

int mfl_meanfield_forward_noalloc(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                  const float *affine_w, const float *affine_b, const float *compat, float *logq,
                                  int B, int C, int H, int W, int k, int iters, mfl_stream_t stream)
{
    const size_t need = mfl_workspace_bytes(MFL_OP_MEANFIELD_FORWARD, B, C, H, W, k, iters);
    if (need > (size_t)MFL_NOALLOC_SCRATCH_BYTES) return MFL_ERR_SCRATCH_LIMIT;
    void *ptr = nullptr;
    const int rc = scratch_pointer(&ptr);
    if (rc) return rc;
    return mfl_meanfield_forward(unary, img, pos_w, col_w, affine_w, affine_b, compat, logq, B, C, H, W, k, iters, ptr,
                                 (size_t)MFL_NOALLOC_SCRATCH_BYTES, stream);
}

int mfl_argmax_labels(const float *logq, unsigned char *labels, int B, int C, int H, int W, mfl_stream_t stream)
{
    int rc = check_common(logq, labels, B, C, H, W, 1);
    if (rc) return rc;
    if (C > 256) return MFL_ERR_INVALID_ARGUMENT;
    const size_t n = (size_t)B * H * W;
    hipLaunchKernelGGL(argmax_kernel, dim3((unsigned)((n + kThreads - 1) / kThreads)), dim3(kThreads), 0,
                       (hipStream_t)stream, logq, labels, B, C, H, W);
    return (int)hipGetLastError();
}

int mfl_iou_counts(const unsigned char *pred, const unsigned char *target, unsigned long long *counts, size_t n, int C,
                   mfl_stream_t stream)
{
    if (!pred || !target || !counts || n == 0 || C <= 0 || C > 256) return MFL_ERR_INVALID_ARGUMENT;
    hipStream_t st = (hipStream_t)stream;
    MFL_CHECK_HIP(hipMemsetAsync(counts, 0, (size_t)3 * C * sizeof(unsigned long long), st));
    // each workgroup's 32-bit histogram counts at most 2^32 - 1 pixels: at least n / 2^31 workgroups
    size_t blocks = (n + (size_t)kThreads * 64 - 1) / ((size_t)kThreads * 64);
    if (blocks < 1) blocks = 1;
    if (blocks > 4096) blocks = 4096;
    if (n / blocks >= (1ull << 31)) return MFL_ERR_TOO_LARGE;
    hipLaunchKernelGGL(iou_counts_kernel, dim3((unsigned)blocks), dim3(kThreads), (size_t)3 * C * 4, st, pred, target, counts, n, C);
    return (int)hipGetLastError();
}

}  // extern "C"

namespace mfl {

bool local_backward_split_supported(int B, int C, int H, int W, int k)
{
    DiagGeom dg;
    return diag_eligible(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, dg) ||
           (diag_block_eligible(B, C, H, W, k, dg) && diag_scatter_chain_ok(C, H, W, k));
}

int local_backward_sums(const float *x, const float *img, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                        mfl_stream_t stream, const float **Pout, const float **Qout)
{
    DiagGeom dg;
    if (!x || !img) return MFL_ERR_INVALID_ARGUMENT;
    if (diag_block_eligible(B, C, H, W, k, dg) && diag_scatter_chain_ok(C, H, W, k)) {
        // more than 33 labels: block by block of output channels
        const BlockBackwardWorkspace ws = block_backward_carve(dg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        float *xp = (float *)(base + ws.f.xp);
        int rc = block_prepare(MFL_OP_LOCAL_BACKWARD, workspace, dg, ws.f, x, img, xp, st);
        if (rc) return rc;
        float *consts = (float *)(base + ws.consts), *P = (float *)(base + ws.psum), *Q = (float *)(base + ws.qsum);
        hipLaunchKernelGGL(diag_backward_consts_kernel, dim3(1), dim3(64), 0, st, consts);
        rc = block_window_products(workspace, dg, ws.f, xp, consts, consts + 1, P, st, 2);
        if (rc) return rc;
        rc = block_window_products(workspace, dg, ws.f, xp, consts + 2, consts + 3, Q, st, 2);
        if (rc) return rc;
        *Pout = P;
        *Qout = Q;
        return MFL_OK;
    }
    if (!diag_eligible(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, dg)) return MFL_ERR_INVALID_ARGUMENT;
    const DiagWorkspace ws = diag_carve(MFL_OP_LOCAL_BACKWARD, dg);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    int rc = diag_prepare(MFL_OP_LOCAL_BACKWARD, workspace, dg, x, img, (float *)(base + ws.xp), (F4 *)(base + ws.ip4),
                          (uint32_t *)(base + ws.table), (uint32_t *)(base + ws.groups), st);
    if (rc) return rc;
    float *consts = (float *)(base + ws.consts), *P = (float *)(base + ws.psum), *Q = (float *)(base + ws.qsum);
    hipLaunchKernelGGL(diag_backward_consts_kernel, dim3(1), dim3(64), 0, st, consts);
    DiagArgs da{};
    da.g = dg;
    diag_lds(dg, da);
    da.xp = (const float *)(base + ws.xp);
    da.ip4 = (const F4 *)(base + ws.ip4);
    da.table = (const uint32_t *)(base + ws.table);
    da.groups = (const uint32_t *)(base + ws.groups);
    da.pos_w = consts; da.col_w = consts + 1; da.y = P;
    rc = dispatch_diag(da, 2, st);
    if (rc) return rc;
    da.pos_w = consts + 2; da.col_w = consts + 3; da.y = Q;
    rc = dispatch_diag(da, 2, st);
    if (rc) return rc;
    *Pout = P;
    *Qout = Q;
    return MFL_OK;
}

int local_backward_apply(const float *x, const float *img, const float *pos_w, const float *col_w, const float *dy, float *dx,
                         float *dparams, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                         mfl_stream_t stream)
{
    DiagGeom dg;
    if (!x || !img || !dy || !dx || !dparams) return MFL_ERR_INVALID_ARGUMENT;
    if (diag_block_eligible(B, C, H, W, k, dg) && diag_scatter_chain_ok(C, H, W, k)) {
        const BlockBackwardWorkspace ws = block_backward_carve(dg);
        if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
        char *base = (char *)workspace;
        hipStream_t st = (hipStream_t)stream;
        float *P = (float *)(base + ws.psum), *Q = (float *)(base + ws.qsum);
        float *dxp = (float *)(base + ws.dxp), *partials = (float *)(base + ws.partials);
        const size_t n = (size_t)B * C * dg.L, cs = diag_copy_stride(dg);
        MFL_CHECK_HIP(hipMemsetAsync(dxp, 0, cs * 2 * k * 4, st));
        DiagArgs da{};
        diag_lds(dg, da);
        da.xp = (const float *)(base + ws.f.xp);
        da.ip4 = (const F4 *)(base + ws.f.ip4);
        da.groups = (const uint32_t *)(base + ws.f.groups);
        da.pos_w = pos_w; da.col_w = col_w; da.y = nullptr;
        da.unary = dy;
        da.qp_out = dxp;
        da.copy_stride = cs;
        for (int blk = 0; blk < ws.f.nblk; ++blk) {
            da.g = block_geom(dg, blk);
            da.table = (const uint32_t *)(base + ws.f.table + blk * ws.f.table_stride);
            const int rc = dispatch_diag(da, 3, st);
            if (rc) return rc;
        }
        const int blocks = grid_for(n) < kDiagBackwardBlocks ? grid_for(n) : kDiagBackwardBlocks;
        launch_finish_copies(k, blocks, st, dy, x, P, Q, dxp + dg.FS, cs, pos_w, col_w, dx, partials, B * C, H, W, dg.Hp, dg.Wp, dg.p0);
        hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kThreads), 0, st, partials, blocks, dparams);
        return (int)hipGetLastError();
    }
    if (!diag_eligible(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, dg)) return MFL_ERR_INVALID_ARGUMENT;
    const DiagWorkspace ws = diag_carve(MFL_OP_LOCAL_BACKWARD, dg);
    if (!workspace || workspace_bytes < ws.total) return MFL_ERR_WORKSPACE;
    char *base = (char *)workspace;
    hipStream_t st = (hipStream_t)stream;
    float *P = (float *)(base + ws.psum), *Q = (float *)(base + ws.qsum);
    float *dxp = (float *)(base + ws.xp2), *partials = (float *)(base + ws.partials);
    const size_t n = (size_t)B * C * dg.L, cs = diag_copy_stride(dg);
    MFL_CHECK_HIP(hipMemsetAsync(dxp, 0, cs * 2 * k * 4, st));
    DiagArgs da{};
    da.g = dg;
    diag_lds(dg, da);
    da.xp = (const float *)(base + ws.xp);
    da.ip4 = (const F4 *)(base + ws.ip4);
    da.table = (const uint32_t *)(base + ws.table);
    da.groups = (const uint32_t *)(base + ws.groups);
    da.pos_w = pos_w; da.col_w = col_w; da.y = nullptr;
    da.unary = dy;
    da.qp_out = dxp;
    da.copy_stride = cs;
    const int rc = dispatch_diag(da, 3, st);
    if (rc) return rc;
    const int blocks = grid_for(n) < kDiagBackwardBlocks ? grid_for(n) : kDiagBackwardBlocks;
    launch_finish_copies(k, blocks, st, dy, x, P, Q, dxp + dg.FS, cs, pos_w, col_w, dx, partials, B * C, H, W, dg.Hp, dg.Wp, dg.p0);
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(1), dim3(kThreads), 0, st, partials, blocks, dparams);
    return (int)hipGetLastError();
}

// The forward loop with logQ_{t+1} of every iteration kept (for the loop backward, mfl_loop_backward.hip).
int meanfield_forward_keep(const float *unary, const float *img, const float *pos_w, const float *col_w, const float *affine_w,
                           const float *affine_b, const float *compat, float *logq_all, int B, int C, int H, int W, int k,
                           int iters, void *workspace, size_t workspace_bytes, mfl_stream_t stream)
{
    if (!logq_all) return MFL_ERR_INVALID_ARGUMENT;
    return meanfield_forward_impl(unary, img, pos_w, col_w, affine_w, affine_b, compat, nullptr, logq_all, B, C, H, W, k, iters,
                                  workspace, workspace_bytes, stream);
}

}  // namespace mfl
