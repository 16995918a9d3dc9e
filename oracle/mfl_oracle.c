/*
 * CPU restatement (plain C + OpenMP) of the reference's local-window
 * message-passing path.  TEST INFRASTRUCTURE ONLY: it is the checker for the
 * HIP library and the "port" CPU baseline timed by bench.py.  The product
 * package never links, loads or calls it.
 *
 * Reference lines restated (paths relative to the reference checkout):
 *   mflo_color_distance     AdvancedModel.py:529-549  (wrapped_color_distance)
 *   mflo_local_forward      LocalConvolution.py:59-90 (Conv2dLocal.forward)
 *   mflo_local_backward     autograd through LocalConvolution.py:74-90
 *   mflo_combine            AdvancedModel.py:32-35 (BlockMinPooling), :242-244
 *   mflo_meanfield_forward  composition of the above, AdvancedModel.py:216-244
 *
 * Pinned by the .npz files under tests/golden/, which were produced by importing the reference
 * (tests/golden/make_golden.py).
 *
 * Layout: fp32, contiguous NCHW; l = y*W + x; K = k*k taps, t = ky*k + kx;
 * zero padding k/2 before and (k-1)/2 after.  All sums are accumulated in
 * double and rounded once, so the oracle is a tighter statement of the
 * arithmetic than the reference's own fp32 evaluation order.
 *
 * The scramble (LocalConvolution.py:80-81): the value that output channel c,
 * pixel l, tap t multiplies is
 *     f = (c*K + t)*L + l;  l' = f / (C*K);  r = f % (C*K);  c' = r / K;  t' = r % K
 *     S = pad(x)[c'][l'/W + t'/k][l'%W + t'%k].
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int mflo_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static inline float img_at(const float *plane, int H, int W, int y, int x)
{
    return (y >= 0 && y < H && x >= 0 && x < W) ? plane[(size_t)y * W + x] : 0.0f;
}

/* squared colour distance between pixel (y,x) and its window tap (ky,kx) */
static inline double cd_one(const float *img, int CH, int H, int W, int k, int y, int x, int ky, int kx)
{
    const int p0 = k / 2;
    double s = 0.0;
    for (int ch = 0; ch < CH; ++ch) {
        const float *pl = img + (size_t)ch * H * W;
        double d = (double)pl[(size_t)y * W + x] - (double)img_at(pl, H, W, y + ky - p0, x + kx - p0);
        s += d * d;
    }
    return s;
}

void mflo_color_distance(const float *img, float *cd, int B, int H, int W, int k)
{
    const int K = k * k;
    const size_t L = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (size_t l = 0; l < L; ++l) {
            const float *im = img + (size_t)b * 3 * L;
            int y = (int)(l / W), x = (int)(l % W);
            for (int t = 0; t < K; ++t)
                cd[((size_t)b * L + l) * K + t] = (float)cd_one(im, 3, H, W, k, y, x, t / k, t % k);
        }
}

/* weight of tap t at pixel l: pd[t]*pw + cd*cw, cd read from a tensor or recomputed from the image */
static inline double weight_at(const float *cd_b, const float *img_b, int H, int W, int k,
                               size_t l, int t, float pw, float cw)
{
    const int p0 = k / 2;
    const int ky = t / k, kx = t % k;
    const double pd = (double)((ky - p0) * (ky - p0) + (kx - p0) * (kx - p0));
    double cdv;
    if (cd_b)
        cdv = (double)cd_b[l * (size_t)(k * k) + t];
    else
        cdv = (double)(float)cd_one(img_b, 3, H, W, k, (int)(l / W), (int)(l % W), ky, kx);
    return pd * (double)pw + cdv * (double)cw;
}

/* out[b,c,l] = sum_t w[b,l,t] * S[b,c,l,t] (window sum before the final "* x") */
static void window_sum(const float *x, const float *cd, const float *img, float pw, float cw,
                       double *out, int B, int C, int H, int W, int k)
{
    const int K = k * k, p0 = k / 2;
    const size_t L = (size_t)H * W, CK = (size_t)C * K;
#pragma omp parallel
    {
        double *w = (double *)malloc(sizeof(double) * K);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (size_t l = 0; l < L; ++l) {
                const float *xb = x + (size_t)b * C * L;
                const float *cdb = cd ? cd + (size_t)b * L * K : NULL;
                const float *imb = img ? img + (size_t)b * 3 * L : NULL;
                for (int t = 0; t < K; ++t)
                    w[t] = weight_at(cdb, imb, H, W, k, l, t, pw, cw);
                for (int c = 0; c < C; ++c) {
                    double acc = 0.0;
                    for (int t = 0; t < K; ++t) {
                        size_t f = ((size_t)c * K + t) * L + l;
                        size_t lp = f / CK, r = f % CK;
                        int cp = (int)(r / K), tp = (int)(r % K);
                        int sy = (int)(lp / W) + tp / k - p0, sx = (int)(lp % W) + tp % k - p0;
                        acc += w[t] * (double)img_at(xb + (size_t)cp * L, H, W, sy, sx);
                    }
                    out[((size_t)b * C + c) * L + l] = acc;
                }
            }
        free(w);
    }
}

void mflo_local_forward(const float *x, const float *cd, const float *img, float pw, float cw,
                        float *y, int B, int C, int H, int W, int k)
{
    const size_t n = (size_t)B * C * H * W;
    double *out = (double *)malloc(sizeof(double) * n);
    window_sum(x, cd, img, pw, cw, out, B, C, H, W, k);
#pragma omp parallel for schedule(static)
    for (size_t i = 0; i < n; ++i)
        y[i] = (float)(out[i] * (double)x[i]);
    free(out);
}

/*
 * Backward as a gather (the adjoint of a gather is a scatter; here every
 * element of pad(x) enumerates its K readers through the inverse index map):
 *   dx[c',y,x] = dy*out + sum_{ky,kx : pixel (y+p0-ky, x+p0-kx) inside}
 *                  D[c,l] * w[l,t],  (c,t,l) decoded from f = l'*C*K + c'*K + t'
 * with D = dy * x.  dparams[0] = d position_weight, dparams[1] = d color_weight.
 */
void mflo_local_backward(const float *x, const float *cd, const float *img, float pw, float cw,
                         const float *dy, float *dx, double *dparams, int B, int C, int H, int W, int k)
{
    const int K = k * k, p0 = k / 2;
    const size_t L = (size_t)H * W, CK = (size_t)C * K, n = (size_t)B * C * L;
    double *out = (double *)malloc(sizeof(double) * n);
    window_sum(x, cd, img, pw, cw, out, B, C, H, W, k);

#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (size_t e = 0; e < (size_t)C * L; ++e) {
            const int cp = (int)(e / L);
            const size_t pix = e % L;
            const int y = (int)(pix / W), xx = (int)(pix % W);
            const size_t base = (size_t)b * C * L;
            const float *cdb = cd ? cd + (size_t)b * L * K : NULL;
            const float *imb = img ? img + (size_t)b * 3 * L : NULL;
            double acc = (double)dy[base + e] * out[base + e];
            for (int tp = 0; tp < K; ++tp) {
                int ys = y + p0 - tp / k, xs = xx + p0 - tp % k;      /* reader pixel l' */
                if (ys < 0 || ys >= H || xs < 0 || xs >= W) continue;
                size_t f = ((size_t)ys * W + xs) * CK + (size_t)cp * K + tp;
                size_t j = f / L, l = f % L;
                int c = (int)(j / K), t = (int)(j % K);
                double D = (double)dy[base + (size_t)c * L + l] * (double)x[base + (size_t)c * L + l];
                acc += D * weight_at(cdb, imb, H, W, k, l, t, pw, cw);
            }
            dx[base + e] = (float)acc;
        }

    double dpw = 0.0, dcw = 0.0;
#pragma omp parallel for collapse(2) schedule(static) reduction(+ : dpw, dcw)
    for (int b = 0; b < B; ++b)
        for (size_t l = 0; l < L; ++l) {
            const float *xb = x + (size_t)b * C * L;
            const float *cdb = cd ? cd + (size_t)b * L * K : NULL;
            const float *imb = img ? img + (size_t)b * 3 * L : NULL;
            for (int t = 0; t < K; ++t) {
                const int ky = t / k, kx = t % k;
                const double pd = (double)((ky - p0) * (ky - p0) + (kx - p0) * (kx - p0));
                const double cdv = cdb ? (double)cdb[l * K + t]
                                       : (double)(float)cd_one(imb, 3, H, W, k, (int)(l / W), (int)(l % W), ky, kx);
                double u = 0.0;
                for (int c = 0; c < C; ++c) {
                    size_t f = ((size_t)c * K + t) * L + l;
                    size_t lp = f / CK, r = f % CK;
                    int cp = (int)(r / K), tp = (int)(r % K);
                    int sy = (int)(lp / W) + tp / k - p0, sx = (int)(lp % W) + tp % k - p0;
                    double D = (double)dy[(size_t)b * C * L + (size_t)c * L + l] * (double)xb[(size_t)c * L + l];
                    u += D * (double)img_at(xb + (size_t)cp * L, H, W, sy, sx);
                }
                dpw += pd * u;
                dcw += cdv * u;
            }
        }
    dparams[0] = dpw;
    dparams[1] = dcw;
    free(out);
}

/* in-place log-softmax over n doubles with stride 1 */
static inline void log_softmax_d(double *v, int n)
{
    double m = v[0];
    for (int i = 1; i < n; ++i) m = v[i] > m ? v[i] : m;
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += exp(v[i] - m);
    const double lse = m + log(s);
    for (int i = 0; i < n; ++i) v[i] -= lse;
}

/*
 * logq = LogSoftmax_c(log_softmax_c(unary) - min_m smooth[b, m*C + c]),
 * smooth is [B, M*C, H, W] (M = 1: plain subtraction).  q (optional) = exp(logq).
 */
void mflo_combine(const float *unary, const float *smooth, float *logq, float *q,
                  int B, int C, int H, int W, int mixtures)
{
    const size_t L = (size_t)H * W;
#pragma omp parallel
    {
        double *u = (double *)malloc(sizeof(double) * C);
#pragma omp for collapse(2) schedule(static)
        for (int b = 0; b < B; ++b)
            for (size_t l = 0; l < L; ++l) {
                for (int c = 0; c < C; ++c) u[c] = (double)unary[((size_t)b * C + c) * L + l];
                log_softmax_d(u, C);
                for (int c = 0; c < C; ++c) {
                    float mn = smooth[((size_t)b * mixtures * C + c) * L + l];
                    for (int m = 1; m < mixtures; ++m) {
                        float v = smooth[((size_t)b * mixtures * C + (size_t)m * C + c) * L + l];
                        mn = v < mn ? v : mn;
                    }
                    u[c] -= (double)mn;
                }
                log_softmax_d(u, C);
                for (int c = 0; c < C; ++c) {
                    logq[((size_t)b * C + c) * L + l] = (float)u[c];
                    if (q) q[((size_t)b * C + c) * L + l] = (float)exp(u[c]);
                }
            }
        free(u);
    }
}

/*
 * Mean-field loop.  Q_0 = unary; per iteration
 *   Msg = x * window_sum(Q_t)          (local_forward, weights from the image)
 *   Msg = aw * Msg + ab                (optional per-pixel affine, [C,H,W] each)
 *   P   = compat @ Msg                 (optional C x C matrix, row-major [c][c'])
 *   logQ = LogSoftmax(log_softmax(unary) - P);  Q_{t+1} = exp(logQ) rounded to fp32
 * Q_t is held in fp32 between iterations, as the device path stores it.
 */
void mflo_meanfield_forward(const float *unary, const float *img, float pw, float cw,
                            const float *aw, const float *ab, const float *compat,
                            float *logq, int B, int C, int H, int W, int k, int iters)
{
    const size_t L = (size_t)H * W, n = (size_t)B * C * L;
    float *q = (float *)malloc(sizeof(float) * n);
    double *out = (double *)malloc(sizeof(double) * n);
    memcpy(q, unary, sizeof(float) * n);
    for (int it = 0; it < iters; ++it) {
        window_sum(q, NULL, img, pw, cw, out, B, C, H, W, k);
#pragma omp parallel
        {
            double *m = (double *)malloc(sizeof(double) * 2 * C);
            double *u = m + C;
#pragma omp for collapse(2) schedule(static)
            for (int b = 0; b < B; ++b)
                for (size_t l = 0; l < L; ++l) {
                    for (int c = 0; c < C; ++c) {
                        size_t i = ((size_t)b * C + c) * L + l;
                        double v = out[i] * (double)q[i];
                        if (aw) v = (double)aw[(size_t)c * L + l] * v;
                        if (ab) v += (double)ab[(size_t)c * L + l];
                        m[c] = v;
                        u[c] = (double)unary[i];
                    }
                    log_softmax_d(u, C);
                    for (int c = 0; c < C; ++c) {
                        double p = m[c];
                        if (compat) {
                            p = 0.0;
                            for (int c2 = 0; c2 < C; ++c2) p += (double)compat[(size_t)c * C + c2] * m[c2];
                        }
                        u[c] -= p;
                    }
                    log_softmax_d(u, C);
                    for (int c = 0; c < C; ++c)
                        logq[((size_t)b * C + c) * L + l] = (float)u[c];
                }
            free(m);
        }
        /* write Q_{t+1} only after every pixel of this iteration has read Q_t */
#pragma omp parallel for schedule(static)
        for (size_t i = 0; i < n; ++i) q[i] = (float)exp((double)logq[i]);
    }
    free(out);
    free(q);
}
