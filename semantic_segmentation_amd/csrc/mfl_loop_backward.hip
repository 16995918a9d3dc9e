// Backward of the K-iteration mean-field loop (C ABI entry point mfl_meanfield_backward): what autograd
// derives when loss.backward() (AdvancedModel.py:299) runs through the composition
//   M_t = Conv2dLocal(Q_t)                LocalConvolution.py:59-90
//   A_t = aw * M_t + ab                    AdvancedModel.py:52-56, 223
//   P_t = compat . A_t                     AdvancedModel.py:143-144 with a 1x1 kernel
//   logQ_{t+1} = LogSoftmax(log_softmax(U) - P_t),  Q_{t+1} = exp(logQ_{t+1}),  Q_0 = U    AdvancedModel.py:222, 242-244
// iterated K times.
//
// Scheme: recompute with checkpoints.  The forward loop runs once more with logQ_{t+1} of every iteration
// kept; then, for t = K-1 .. 0:
//   X_t  = Q_t (t = 0: U; else exp of the kept logQ_t)                         exp kernel
//   M_t  = local forward of X_t                                                mfl_local_forward
//   pre  : g = incoming gradient of logQ_{t+1} (t = K-1: the caller's; else dQ_{t+1} * Q_{t+1})
//          dz = g - Q_{t+1} * sum_c g;  dls += dz;  dP = -dz;  dA = compat^T . dP;  dM = aw * dA
//          daw += sum_b dA * M_t;  dab += sum_b dA;  dcompat += sum_{b,l} dP (x) A_t    loop_pre_kernel + reduce
//   (dQ_t, d pos_w, d col_w) = local backward of (X_t, dM)                     mfl_local_backward
// and finally dU = dQ_0 + dls - softmax(U) * sum_c dls.
// Every sum over pixels or images has a fixed order (no atomics here): apart from the scatter inside
// mfl_local_backward the result is reproducible bit for bit.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfl.h"

#define MFL_CHECK_HIP(expr)                          \
    do {                                             \
        const hipError_t e_ = (expr);                \
        if (e_ != hipSuccess) return (int)e_;        \
    } while (0)

namespace {

constexpr int kPix = 64;          // pixels per workgroup of the pre kernel (one wave)
constexpr int kPitch = kPix + 1;  // LDS pitch of a channel row: rows read across channels stay conflict-free
constexpr int kMaxC = 32;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

__global__ void loop_exp_kernel(const float *__restrict__ logq, float *__restrict__ q, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        q[i] = __expf(logq[i]);
}

// One workgroup (one wave) per 64 consecutive pixels of the image plane, looping over the batch so that
// the sums over images (daw, dab) and over the pixels of the tile (dcompat) have a fixed order.
//   g_in:    t = K-1: dL/dlogQ_K of the caller;  else dQ_{t+1} of the last local backward (from_dq)
//   logq1:   logQ_{t+1}
//   m:       M_t (local forward of Q_t), or NULL: then M_t = Q_t * (pos_w * P + col_w * Q) from the plain window
//            sums P, Q of the local backward's first phase (xq = Q_t)
//   dm:      out, gradient of M_t
//   dls:     running sum of dz (first: overwritten)
//   daw/dab: running sums (first: overwritten), [C][L]; NULL without the affine
//   part:    out, this workgroup's C x C partial of dcompat (NULL without the matrix)
__global__ void __launch_bounds__(kPix)
loop_pre_kernel(const float *__restrict__ g_in, int from_dq, const float *__restrict__ logq1, const float *__restrict__ m,
                const float *__restrict__ xq, const float *__restrict__ psum, const float *__restrict__ qsum,
                const float *__restrict__ pos_w, const float *__restrict__ col_w, const float *__restrict__ aw, const float *__restrict__ ab, const float *__restrict__ compat,
                float *__restrict__ dm, float *__restrict__ dls, float *__restrict__ daw, float *__restrict__ dab,
                float *__restrict__ part, int first, int B, int C, int L)
{
    extern __shared__ float lds[];
    float *dp = lds;                       // [C][kPitch]  dP of the current image
    float *at = dp + C * kPitch;           // [C][kPitch]  A_t of the current image
    float *sw = at + C * kPitch;           // [C][kPitch]  sum_b dA * M
    float *sb = sw + C * kPitch;           // [C][kPitch]  sum_b dA
    float *cacc = sb + C * kPitch;         // [C*C]        sum dP (x) A
    float *cm = cacc + C * C;              // [C*C]        the label matrix (read once per workgroup)
    const int lane = threadIdx.x;
    const int l = blockIdx.x * kPix + lane;
    const bool ok = l < L;
    const float pw = m ? 0.f : *pos_w, cw = m ? 0.f : *col_w;
    auto message = [&](size_t o) { return m ? m[o] : xq[o] * fmaf(pw, psum[o], cw * qsum[o]); };
    for (int c = 0; c < C; ++c) { sw[c * kPitch + lane] = 0.f; sb[c * kPitch + lane] = 0.f; }
    for (int i = lane; i < C * C; i += kPix) {
        cacc[i] = 0.f;
        cm[i] = compat ? compat[i] : 0.f;
    }
    __syncthreads();
    for (int b = 0; b < B; ++b) {
        const size_t base = (size_t)b * C * L + l;
        float sg = 0.f;
        for (int c = 0; c < C; ++c) {
            float g = 0.f, q = 0.f, a = 0.f;
            if (ok) {
                q = __expf(logq1[base + (size_t)c * L]);
                g = g_in[base + (size_t)c * L];
                if (from_dq) g *= q;
                const float mv = message(base + (size_t)c * L);
                a = aw ? fmaf(aw[(size_t)c * L + l], mv, ab[(size_t)c * L + l]) : mv;
            }
            dp[c * kPitch + lane] = g;      // g for now
            at[c * kPitch + lane] = a;
            sg += g;
        }
        for (int c = 0; c < C; ++c) {
            float dz = 0.f;
            if (ok) {
                const float q = __expf(logq1[base + (size_t)c * L]);
                dz = dp[c * kPitch + lane] - q * sg;
                dls[base + (size_t)c * L] = first ? dz : dls[base + (size_t)c * L] + dz;
            }
            dp[c * kPitch + lane] = -dz;
        }
        __syncthreads();
        // dA_j = sum_i compat[i][j] * dP_i (this thread's pixel); dM_j = aw_j * dA_j
        for (int j = 0; j < C; ++j) {
            float da;
            if (compat) {
                da = 0.f;
                for (int i = 0; i < C; ++i) da = fmaf(cm[i * C + j], dp[i * kPitch + lane], da);
            } else {
                da = dp[j * kPitch + lane];
            }
            if (ok) {
                float dmv = da;
                if (aw) {
                    const float mv = message(base + (size_t)j * L);
                    sw[j * kPitch + lane] = fmaf(da, mv, sw[j * kPitch + lane]);
                    sb[j * kPitch + lane] += da;
                    dmv = aw[(size_t)j * L + l] * da;
                }
                dm[base + (size_t)j * L] = dmv;
            }
        }
        if (part) {
            // pair (i, j) of the matrix: sum over the tile's pixels (rows of pixels past L hold zeros)
            for (int idx = lane; idx < C * C; idx += kPix) {
                const int i = idx / C, j = idx - i * C;
                float s = 0.f;
                for (int px = 0; px < kPix; ++px) s = fmaf(dp[i * kPitch + px], at[j * kPitch + px], s);
                cacc[idx] += s;
            }
        }
        __syncthreads();
    }
    if (ok && daw) {
        for (int c = 0; c < C; ++c) {
            const size_t o = (size_t)c * L + l;
            daw[o] = first ? sw[c * kPitch + lane] : daw[o] + sw[c * kPitch + lane];
            dab[o] = first ? sb[c * kPitch + lane] : dab[o] + sb[c * kPitch + lane];
        }
    }
    if (part)
        for (int i = lane; i < C * C; i += kPix) part[(size_t)blockIdx.x * C * C + i] = cacc[i];
}

// dcompat[i][j] (+)= sum over the workgroups' partials: one workgroup per matrix element, every thread sums a
// fixed strided subset of the partials, then a fixed tree in LDS (the same order in every run).
__global__ void __launch_bounds__(256)
loop_compat_reduce_kernel(const float *__restrict__ part, int nblk, int CC, float *__restrict__ dcompat, int first)
{
    __shared__ float sh[256];
    const int idx = blockIdx.x;
    float s = 0.f;
    for (int b = threadIdx.x; b < nblk; b += 256) s += part[(size_t)b * CC + idx];
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) sh[threadIdx.x] += sh[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) dcompat[idx] = first ? sh[0] : dcompat[idx] + sh[0];
}

// dU = dQ_0 + dls - softmax(U) * sum_c dls; one thread per pixel.  dparams = sum of the per-iteration pairs.
__global__ void loop_final_kernel(const float *__restrict__ unary, const float *__restrict__ dq0, const float *__restrict__ dls,
                                  float *__restrict__ du, const float *__restrict__ dpar_slots, float *__restrict__ dparams,
                                  int iters, int B, int C, int L)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {
        float a = 0.f, b = 0.f;
        for (int t = iters - 1; t >= 0; --t) { a += dpar_slots[2 * t]; b += dpar_slots[2 * t + 1]; }
        dparams[0] = a;
        dparams[1] = b;
    }
    if (i >= (size_t)B * L) return;
    const size_t b = i / L, l = i - b * L;
    const size_t base = b * C * L + l;
    float mx = -INFINITY, sd = 0.f;
    for (int c = 0; c < C; ++c) {
        mx = fmaxf(mx, unary[base + (size_t)c * L]);
        sd += dls[base + (size_t)c * L];
    }
    float se = 0.f;
    for (int c = 0; c < C; ++c) se += __expf(unary[base + (size_t)c * L] - mx);
    const float rs = 1.f / se;
    for (int c = 0; c < C; ++c) {
        const size_t o = base + (size_t)c * L;
        du[o] = dq0[o] + dls[o] - __expf(unary[o] - mx) * rs * sd;
    }
}

struct LoopWorkspace {
    size_t logq_all, x, m, dm, dx, dls, dpar, part, sub, sub_bytes, total;
    int nblk;
};

LoopWorkspace loop_carve(int B, int C, int H, int W, int k, int iters)
{
    LoopWorkspace w{};
    const size_t n = (size_t)B * C * H * W * 4;
    const int L = H * W;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    w.logq_all = take(n * (size_t)iters);
    w.x = take(n);
    w.m = take(n);
    w.dm = take(n);
    w.dx = take(n);
    w.dls = take(n);
    w.dpar = take((size_t)iters * 2 * 4);
    w.nblk = (L + kPix - 1) / kPix;
    w.part = take((size_t)w.nblk * C * C * 4);
    size_t sub = mfl_workspace_bytes(MFL_OP_MEANFIELD_FORWARD, B, C, H, W, k, iters);
    const size_t f = mfl_workspace_bytes(MFL_OP_LOCAL_FORWARD, B, C, H, W, k, 1);
    const size_t bw = mfl_workspace_bytes(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, 1);
    if (f > sub) sub = f;
    if (bw > sub) sub = bw;
    w.sub_bytes = sub;
    w.sub = take(sub);
    w.total = off;
    return w;
}

}  // namespace

namespace mfl {

// (mfl_api.hip: the diagonal local backward in two phases)
bool local_backward_split_supported(int B, int C, int H, int W, int k);
int local_backward_sums(const float *x, const float *img, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                        mfl_stream_t stream, const float **P, const float **Q);
int local_backward_apply(const float *x, const float *img, const float *pos_w, const float *col_w, const float *dy, float *dx,
                         float *dparams, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                         mfl_stream_t stream);

// (declared in mfl_api.hip: the forward loop that keeps every iterate)
int meanfield_forward_keep(const float *unary, const float *img, const float *pos_w, const float *col_w, const float *affine_w,
                           const float *affine_b, const float *compat, float *logq_all, int B, int C, int H, int W, int k,
                           int iters, void *workspace, size_t workspace_bytes, mfl_stream_t stream);

size_t loop_backward_workspace_bytes(int B, int C, int H, int W, int k, int iters)
{
    if (B <= 0 || C <= 0 || C > kMaxC || H <= 0 || W <= 0 || k <= 0 || iters <= 0) return 0;
    if (mfl_workspace_bytes(MFL_OP_LOCAL_BACKWARD, B, C, H, W, k, 1) == 0) return 0;
    return loop_carve(B, C, H, W, k, iters).total;
}

}  // namespace mfl

extern "C" int mfl_meanfield_backward(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                      const float *affine_w, const float *affine_b, const float *compat,
                                      const float *logq_iterates, const float *grad_logq, float *d_unary, float *dparams,
                                      float *d_affine_w,
                                      float *d_affine_b, float *d_compat, int B, int C, int H, int W, int k, int iters,
                                      void *workspace, size_t workspace_bytes, mfl_stream_t stream)
{
    if (!unary || !img || !pos_w || !col_w || !grad_logq || !d_unary || !dparams) return MFL_ERR_INVALID_ARGUMENT;
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || k <= 0 || iters <= 0) return MFL_ERR_INVALID_ARGUMENT;
    if ((affine_w == nullptr) != (affine_b == nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if ((affine_w != nullptr) != (d_affine_w != nullptr) || (affine_w != nullptr) != (d_affine_b != nullptr))
        return MFL_ERR_INVALID_ARGUMENT;
    if ((compat != nullptr) != (d_compat != nullptr)) return MFL_ERR_INVALID_ARGUMENT;
    if (C > kMaxC) return MFL_ERR_INVALID_ARGUMENT;           // the pre kernel holds four C x 65 tiles and C x C sums in LDS
    if ((long long)B * C * H * W >= (1LL << 31)) return MFL_ERR_TOO_LARGE;
    const size_t need = mfl::loop_backward_workspace_bytes(B, C, H, W, k, iters);
    if (need == 0) return MFL_ERR_INVALID_ARGUMENT;
    if (!workspace || workspace_bytes < need) return MFL_ERR_WORKSPACE;
    const LoopWorkspace ws = loop_carve(B, C, H, W, k, iters);
    char *base = (char *)workspace;
    const float *logq_all = logq_iterates ? logq_iterates : (const float *)(base + ws.logq_all);
    float *x = (float *)(base + ws.x), *m = (float *)(base + ws.m);
    float *dm = (float *)(base + ws.dm), *dx = (float *)(base + ws.dx), *dls = (float *)(base + ws.dls);
    float *dpar = (float *)(base + ws.dpar), *part = (float *)(base + ws.part);
    void *sub = base + ws.sub;
    hipStream_t st = (hipStream_t)stream;
    const int L = H * W;
    const size_t n = (size_t)B * C * L;

    int rc = 0;
    if (!logq_iterates) {
        // no iterates from the caller (mfl_meanfield_forward_iterates): the forward loop runs again and keeps them
        rc = mfl::meanfield_forward_keep(unary, img, pos_w, col_w, affine_w, affine_b, compat, (float *)(base + ws.logq_all), B, C, H,
                                         W, k, iters, sub, ws.sub_bytes, stream);
        if (rc) return rc;
    }
    const size_t lds = ((size_t)4 * C * kPitch + (size_t)2 * C * C) * 4;
    const bool split = mfl::local_backward_split_supported(B, C, H, W, k);
    const unsigned egrid = (unsigned)((n + 255) / 256 < 65535 * 8 ? (n + 255) / 256 : 65535 * 8);
    for (int t = iters - 1; t >= 0; --t) {
        const float *xt = unary;
        if (t > 0) {
            hipLaunchKernelGGL(loop_exp_kernel, dim3(egrid), dim3(256), 0, st, logq_all + (size_t)(t - 1) * n, x, n);
            xt = x;
        }
        // Where the diagonal kernels serve the local backward, its first phase (the plain window sums P, Q, which do
        // not depend on the incoming gradient) also yields M_t = X_t * (pos_w * P + col_w * Q): no separate local forward.
        const float *psum = nullptr, *qsum = nullptr;
        if (split) {
            rc = mfl::local_backward_sums(xt, img, B, C, H, W, k, sub, ws.sub_bytes, stream, &psum, &qsum);
        } else {
            rc = mfl_local_forward(xt, img, nullptr, pos_w, col_w, m, B, C, H, W, k, sub, ws.sub_bytes, stream);
        }
        if (rc) return rc;
        const int first = t == iters - 1;
        hipLaunchKernelGGL(loop_pre_kernel, dim3(ws.nblk), dim3(kPix), lds, st, first ? grad_logq : dx, first ? 0 : 1,
                           logq_all + (size_t)t * n, split ? nullptr : m, xt, psum, qsum, pos_w, col_w, affine_w, affine_b, compat, dm,
                           dls, d_affine_w, d_affine_b, compat ? part : nullptr, first, B, C, L);
        if (compat)
            hipLaunchKernelGGL(loop_compat_reduce_kernel, dim3(C * C), dim3(256), 0, st, part, ws.nblk, C * C, d_compat, first);
        MFL_CHECK_HIP(hipGetLastError());
        if (split) {
            rc = mfl::local_backward_apply(xt, img, pos_w, col_w, dm, dx, dpar + 2 * t, B, C, H, W, k, sub, ws.sub_bytes, stream);
        } else {
            rc = mfl_local_backward(xt, img, nullptr, pos_w, col_w, dm, dx, dpar + 2 * t, nullptr, B, C, H, W, k, sub, ws.sub_bytes,
                                    stream);
        }
        if (rc) return rc;
    }
    hipLaunchKernelGGL(loop_final_kernel, dim3((unsigned)(((size_t)B * L + 255) / 256)), dim3(256), 0, st, unary, dx, dls, d_unary,
                       dpar, dparams, iters, B, C, L);
    return (int)hipGetLastError();
}
