/*
 * mfl.h - C ABI of the MI355X (gfx950) local-window message-passing library.
 *
 * "mfl" = mean-field, local window.  The library replaces, for one hot path,
 * what the reference computes with stock PyTorch ops:
 *
 *   Conv2dLocal.forward                     LocalConvolution.py:59-90
 *   DPNModel._color_distance                AdvancedModel.py:520-551
 *   autograd backward of Conv2dLocal        driven by AdvancedModel.py:299
 *   BlockMinPooling + unary/smoothness      AdvancedModel.py:25-35, 242-244
 *   combine + LogSoftmax
 *   LinearActivation (per-pixel affine)     AdvancedModel.py:38-56
 *
 * The reference has no FFI of its own (it is pure Python); each entry point
 * below names the Python symbol whose arithmetic it takes over; the binding a
 * maintainer adds on the reference side is a ctypes one
 * (semantic_segmentation_amd/_lib.py is such a binding).
 *
 * Conventions
 *   - plain C, no torch types; every pointer is a BORROWED device pointer that
 *     must stay valid until the work queued on `stream` has finished;
 *   - tensors are fp32, contiguous NCHW: x[B][C][H][W]; pixels are flattened
 *     row-major (l = y*W + x); the window is k x k with K = k*k taps flattened
 *     ky*k + kx; zero padding is k/2 before and (k-1)/2 after on both axes;
 *   - the library allocates nothing and never synchronises the device; scratch
 *     memory is a caller-provided workspace whose size mfl_workspace_bytes()
 *     reports; kernels are queued on `stream` (a hipStream_t passed as void*;
 *     NULL = the default stream) of the device that is current at the call;
 *   - potential parameters (position_weight, color_weight) are passed as device
 *     pointers to one float each, so no host read-back is needed;
 *   - every function returns 0 on success, a negative MFL_ERR_* code for a
 *     rejected argument, or a positive hipError_t from the runtime.  Nothing
 *     throws or aborts across this boundary.
 */
#ifndef MFL_H
#define MFL_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MFL_VERSION 100 /* 0.1.0 */

#define MFL_OK 0
#define MFL_ERR_INVALID_ARGUMENT (-1) /* null pointer, non-positive size, unsupported combination */
#define MFL_ERR_WORKSPACE (-2)        /* workspace missing or smaller than mfl_workspace_bytes() */
#define MFL_ERR_TOO_LARGE (-3)        /* a 32-bit index inside the kernels would overflow */
#define MFL_ERR_SCRATCH_LIMIT (-4)    /* mfl_meanfield_forward_noalloc: needs more than MFL_NOALLOC_SCRATCH_BYTES */

/* size of the library's built-in scratch buffer (one per GPU, held until the process ends) */
#define MFL_NOALLOC_SCRATCH_BYTES (64u << 20)

/* operations for mfl_workspace_bytes() */
#define MFL_OP_LOCAL_FORWARD 1
#define MFL_OP_LOCAL_BACKWARD 2
#define MFL_OP_MEANFIELD_FORWARD 3
#define MFL_OP_MEANFIELD_BACKWARD 4

typedef void *mfl_stream_t;

/* Library version (MFL_VERSION of the build). */
int mfl_version(void);

/* Static description of a return code (never NULL). */
const char *mfl_error_string(int code);

/*
 * Bytes of device workspace the given operation needs for this problem size
 * (0 if the operation needs none or the arguments are invalid).  `iters` is
 * only read for MFL_OP_MEANFIELD_BACKWARD.
 */
size_t mfl_workspace_bytes(int op, int B, int C, int H, int W, int k, int iters);

/*
 * Colour-distance tensor, materialised (the fused paths never need it; it exists
 * for callers that want the tensor itself and for tests).
 *   cd[b][l][t] = sum_ch (img[b][ch][y][x] - pad(img)[b][ch][y+ky][x+kx])^2
 * Replaces DPNModel._color_distance / wrapped_color_distance,
 * AdvancedModel.py:520-551.   img: [B][3][H][W]   cd: [B][H*W][k*k]
 */
int mfl_color_distance(const float *img, float *cd, int B, int H, int W, int k, mfl_stream_t stream);

/*
 * y = x * sum_t (pd[t]*pos_w + cdist[l][t]*col_w) * S[c][l][t], where S is the
 * reference's permuted/reshaped im2col of zero-padded x and pd the squared tap
 * offsets.  Replaces Conv2dLocal.forward, LocalConvolution.py:59-90, fused with
 * the colour distance when `img` is given.
 * Exactly one of `img` ([B][3][H][W]; the distance is recomputed on chip) and
 * `cd` ([B][H*W][k*k]; read from memory) must be non-NULL.
 *   x, y: [B][C][H][W]   pos_w, col_w: device pointers to 1 float
 */
int mfl_local_forward(const float *x, const float *img, const float *cd, const float *pos_w, const float *col_w,
                      float *y, int B, int C, int H, int W, int k, void *workspace, size_t workspace_bytes,
                      mfl_stream_t stream);

/*
 * Gradients of mfl_local_forward for an incoming dy (what autograd derives from
 * LocalConvolution.py:74-90 when AdvancedModel.py:299 calls loss.backward()).
 *   dx:      [B][C][H][W]
 *   dparams: 2 floats, overwritten with {d position_weight, d color_weight}
 *   dcd:     optional [B][H*W][k*k] gradient w.r.t. the colour-distance tensor
 *            (NULL to skip; only allowed with `cd`)
 * dx is reproducible bit for bit with `img` on two families of shapes:
 *  - the diagonal kernels' shapes (images at least 64 + k pixels wide, odd window
 *    3..11, at least k * C * k * k pixels; more than 33 labels block by block of
 *    output channels): what the window products send to an element of pad(x) is
 *    stored into one of 2k copies of the gradient planes in which that element
 *    has a single writer (two, adding atomically onto zero, near the ends of the
 *    pixel range), and the copies are added in a fixed order.  The workspace
 *    holds those copies: 2k * B * C * (H + k - 1) * (W + k - 1) floats.
 *  - large windows over small images (at least 16 x 16 taps on at most 64 K
 *    pixels: the reference model's own 50 x 50 window on 64 x 64): summed in
 *    64-bit fixed point with integer atomics.
 * The other paths (a `cd` tensor, narrow images, even windows on large images)
 * use float atomics, whose summation order and therefore last bits can differ
 * between runs.
 */
int mfl_local_backward(const float *x, const float *img, const float *cd, const float *pos_w, const float *col_w,
                       const float *dy, float *dx, float *dparams, float *dcd, int B, int C, int H, int W, int k,
                       void *workspace, size_t workspace_bytes, mfl_stream_t stream);

/*
 * logq = LogSoftmax_c( log_softmax_c(unary) - min_m smooth[b][m*C + c] ).
 * Replaces BlockMinPooling.forward (AdvancedModel.py:32-35; pass mixtures = 1
 * for a plain smoothness tensor) and the combine + final LogSoftmax,
 * AdvancedModel.py:242-244.
 *   unary, logq: [B][C][H][W]   smooth: [B][mixtures*C][H][W]
 *   q: optional exp(logq), same shape (NULL to skip)
 */
int mfl_combine_forward(const float *unary, const float *smooth, int mixtures, float *logq, float *q, int B, int C,
                        int H, int W, mfl_stream_t stream);

/*
 * `iters` fused mean-field updates.  Q_0 = unary (the reference feeds the raw
 * unary to its local layer, AdvancedModel.py:222); each iteration computes
 *   M    = Conv2dLocal(Q_t, colour distance of img)        LocalConvolution.py:59-90
 *   M    = affine_w * M + affine_b        (optional)        AdvancedModel.py:52-56, 223
 *   P    = compat . M                     (optional, CxC)   AdvancedModel.py:143-144 with a 1x1 kernel
 *   logQ = LogSoftmax(log_softmax(unary) - P)               AdvancedModel.py:242-244
 *   Q_{t+1} = exp(logQ)
 * and the last logQ is written to `logq`.  iters = 1 without affine/compat is the
 * reference's phase-4 data flow with the label-context block replaced by identity.
 *   unary, logq: [B][C][H][W]   img: [B][3][H][W]
 *   affine_w, affine_b: [C][H][W] or NULL (both or neither)
 *   compat: [C][C] row-major (P[c] = sum_c' compat[c][c'] * M[c']) or NULL
 */
int mfl_meanfield_forward(const float *unary, const float *img, const float *pos_w, const float *col_w,
                          const float *affine_w, const float *affine_b, const float *compat, float *logq, int B,
                          int C, int H, int W, int k, int iters, void *workspace, size_t workspace_bytes,
                          mfl_stream_t stream);

/*
 * mfl_meanfield_forward that keeps every iterate: logq_iterates[t] = logQ_{t+1} for
 * t = 0 .. iters-1, each [B][C][H][W] (the last one is what mfl_meanfield_forward
 * returns).  For training: handing them to mfl_meanfield_backward saves it the
 * recomputation of the forward loop.  Same workspace as MFL_OP_MEANFIELD_FORWARD.
 */
int mfl_meanfield_forward_iterates(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                   const float *affine_w, const float *affine_b, const float *compat, float *logq_iterates,
                                   int B, int C, int H, int W, int k, int iters, void *workspace, size_t workspace_bytes,
                                   mfl_stream_t stream);

/*
 * Gradients of mfl_meanfield_forward for an incoming grad_logq = dL/dlogq: what
 * autograd derives when loss.backward() (AdvancedModel.py:299) runs through the
 * iterated composition AdvancedModel.py:222-244 (see mfl_meanfield_forward).
 * logq_iterates: the [iters][B][C][H][W] output of mfl_meanfield_forward_iterates for
 * the same inputs, or NULL: then the forward loop is recomputed inside the call with
 * every iterate kept in the workspace (mfl_workspace_bytes(MFL_OP_MEANFIELD_BACKWARD,
 * ...) grows with iters either way).
 *   grad_logq, d_unary: [B][C][H][W]
 *   dparams:    2 floats, overwritten with {d position_weight, d color_weight}
 *   d_affine_w, d_affine_b: [C][H][W], overwritten; non-NULL exactly when affine_w is
 *   d_compat:   [C][C], overwritten; non-NULL exactly when compat is
 * C <= 160 (up to 32 labels a pixel's channel vectors stay in one lane's registers;
 * above that the label products run as GEMMs on the fp32 matrix instructions).  All sums
 * over images and pixels outside the local operator's own backward have a fixed order.  The address tables of the nested operator calls
 * are rebuilt once per call (their plan tag lies inside this workspace and is
 * cleared on entry), so the workspace may be any memory, whatever it held before.
 */
int mfl_meanfield_backward(const float *unary, const float *img, const float *pos_w, const float *col_w,
                           const float *affine_w, const float *affine_b, const float *compat, const float *logq_iterates,
                           const float *grad_logq, float *d_unary, float *dparams, float *d_affine_w, float *d_affine_b,
                           float *d_compat,
                           int B, int C, int H, int W, int k, int iters, void *workspace, size_t workspace_bytes,
                           mfl_stream_t stream);

/*
 * mfl_meanfield_forward without a workspace argument, for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.
 * The scratch memory is a buffer of MFL_NOALLOC_SCRATCH_BYTES built into the
 * library's device image; the call allocates and frees nothing.  Problems with
 * mfl_workspace_bytes(MFL_OP_MEANFIELD_FORWARD, ...) above that size are rejected
 * with MFL_ERR_SCRATCH_LIMIT (64 MiB holds one 21-label 512x512 image with a
 * 7x7 window).  All calls on one GPU share that GPU's buffer: calls that overlap
 * in time on one GPU (different streams or threads) race; serialise them.  The
 * buffer is part of the library's code object: every GPU the library runs on
 * holds its MFL_NOALLOC_SCRATCH_BYTES while the library is loaded, whether this
 * entry point is ever called or not.  The
 * first call on a device looks up the buffer's address (set-up; do it once
 * before capturing a graph).  Same arguments, results and return codes as
 * mfl_meanfield_forward otherwise.
 */
int mfl_meanfield_forward_noalloc(const float *unary, const float *img, const float *pos_w, const float *col_w,
                                  const float *affine_w, const float *affine_b, const float *compat, float *logq,
                                  int B, int C, int H, int W, int k, int iters, mfl_stream_t stream);

/*
 * labels[b][y][x] = argmax_c logq[b][c][y][x] (first maximum wins, as
 * torch.argmax does for the reference's prediction_to_image,
 * AdvancedModel.py:420-421).   labels: [B][H][W] uint8, C <= 256.
 */
int mfl_argmax_labels(const float *logq, unsigned char *labels, int B, int C, int H, int W, mfl_stream_t stream);

/*
 * The reference's label-context block and the combine step behind it, fused:
 *   a      = la_w * t + la_b                  local_activation_function,  AdvancedModel.py:138, 223
 *   c      = ga_w * Conv2d(L -> L*M, ks x ks, pad ks/2)(a) + ga_b
 *                                             global_convolution_layer / global_activation_function,
 *                                             AdvancedModel.py:143-145, 232-233
 *   smooth = min over the M mixtures of c     BlockMinPooling, AdvancedModel.py:32-35, 150, 235
 *   logq   = LogSoftmax(log_softmax(unary) - smooth)                       AdvancedModel.py:242-244
 * i.e. DPNModel.forward phases 3 and 4 behind the local convolution.  The convolution runs as an implicit GEMM
 * on the fp32 matrix instructions.
 *   t, unary, smooth, logq: [B][L][H][W]      (smooth, logq: outputs, either may be NULL, not both)
 *   la_w, la_b: [L][H][W] or NULL (both or neither)     ga_w, ga_b: [L*M][H][W] or NULL (both or neither)
 *   conv_w: [L*M][L][ks][ks]   conv_b: [L*M] or NULL
 * L <= 32, M <= 5, ks odd and <= 15 (the reference: L = 21, M = 5, ks = 9).
 * Workspace: mfl_label_context_workspace_bytes().
 */
size_t mfl_label_context_workspace_bytes(int B, int L, int M, int H, int W, int ks);
int mfl_label_context_forward(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                              const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                              float *smooth, float *logq, int B, int L, int M, int H, int W, int ks, void *workspace,
                              size_t workspace_bytes, mfl_stream_t stream);

/*
 * Gradients of mfl_label_context_forward: what autograd derives for the same chain
 * (local_activation_function -> global_convolution_layer -> global_activation_function ->
 * BlockMinPooling -> combine + LogSoftmax, AdvancedModel.py:223, 232-244) when
 * loss.backward() (AdvancedModel.py:299) runs in training phases 3 and 4
 * (AdvancedModel.py:364-381).  The minimum over the mixtures hands its gradient to the
 * arg-min mixture (the first one on ties, as torch.min(dim)).
 *   grad_logq:   dL/dlogq, [B][L][H][W], or NULL
 *   grad_smooth: dL/dsmooth, [B][L][H][W], or NULL (phase 3 ends in log_softmax(smooth),
 *                AdvancedModel.py:237-238: its gradient enters here); not both NULL
 *   d_t, d_unary: [B][L][H][W], overwritten
 *   d_conv_w: [L*M][L][ks][ks], overwritten    d_conv_b: [L*M], non-NULL exactly when conv_b is
 *   d_la_w, d_la_b: [L][H][W], non-NULL exactly when la_w is
 *   d_ga_w, d_ga_b: [L*M][H][W], non-NULL exactly when ga_w is
 * The forward is recomputed inside the call; the data gradient and the weight gradient of the
 * convolution run as GEMMs on the fp32 matrix instructions.  Every sum over images and pixels
 * has a fixed order (no atomics): the gradients are reproducible bit for bit.
 * Workspace: mfl_label_context_backward_workspace_bytes() (0: unsupported or too large).
 */
size_t mfl_label_context_backward_workspace_bytes(int B, int L, int M, int H, int W, int ks);
int mfl_label_context_backward(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                               const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                               const float *grad_logq, const float *grad_smooth, float *d_t, float *d_unary,
                               float *d_la_w, float *d_la_b, float *d_conv_w, float *d_conv_b, float *d_ga_w,
                               float *d_ga_b, int B, int L, int M, int H, int W, int ks, void *workspace,
                               size_t workspace_bytes, mfl_stream_t stream);

/*
 * The same two calls for a caller that keeps what the forward produced (what autograd keeps alive
 * between DPNModel.forward and loss.backward(), AdvancedModel.py:298-299) instead of paying for the
 * forward twice:
 *   mfl_label_context_forward_keep   also writes, per (image, label, pixel), the arg-min mixture
 *                                    (kept_amin, [B][L][H][W] bytes) and the convolution's value there
 *                                    before the second affine (kept_cmin, [B][L][H][W] floats);
 *                                    logq is required, smooth optional
 *   mfl_label_context_backward_kept  takes logq and those two tensors back and skips the recomputation
 *                                    (the padded activations, an operand of the weight gradient, are
 *                                    rebuilt: one elementwise pass)
 * Results equal those of mfl_label_context_forward / mfl_label_context_backward bit for bit.  Same
 * workspaces as the calls above.
 */
int mfl_label_context_forward_keep(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                   const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                   float *smooth, float *logq, unsigned char *kept_amin, float *kept_cmin, int B, int L,
                                   int M, int H, int W, int ks, void *workspace, size_t workspace_bytes,
                                   mfl_stream_t stream);
int mfl_label_context_backward_kept(const float *t, const float *la_w, const float *la_b, const float *conv_w,
                                    const float *conv_b, const float *ga_w, const float *ga_b, const float *unary,
                                    const float *kept_logq, const unsigned char *kept_amin, const float *kept_cmin,
                                    const float *grad_logq, const float *grad_smooth, float *d_t, float *d_unary,
                                    float *d_la_w, float *d_la_b, float *d_conv_w, float *d_conv_b, float *d_ga_w,
                                    float *d_ga_b, int B, int L, int M, int H, int W, int ks, void *workspace,
                                    size_t workspace_bytes, mfl_stream_t stream);

/*
 * Per-class pixel counts behind the reference's intersection-over-union
 * (DPNModel.calc_intersection_over_union, AdvancedModel.py:448-485):
 *   counts[0][c] = #(pred == c and target == c)    intersection
 *   counts[1][c] = #(pred == c)
 *   counts[2][c] = #(target == c)
 * so that IoU_c = counts[0][c] / (counts[1][c] + counts[2][c] - counts[0][c]).
 * counts: [3][C] unsigned 64-bit, overwritten.  pred, target: n labels (uint8, e.g.
 * the output of mfl_argmax_labels over any number of images); labels >= C (255 =
 * void) belong to no class.  Integer sums: the result is exact and order-free.
 */
int mfl_iou_counts(const unsigned char *pred, const unsigned char *target, unsigned long long *counts, size_t n, int C,
                   mfl_stream_t stream);

/*
 * The image preprocessing in front of the path, on the device: what DPNModel._preprocess
 * (AdvancedModel.py:487-498: transforms.Resize((H, W)), ToTensor, Normalize) does per image on the
 * host.  Resize on a PIL image is Pillow's two-pass bilinear resampling of the 8-bit pixels with
 * fixed-point coefficients and an 8-bit intermediate; the caller computes the coefficient tables
 * on the host exactly as Pillow does (semantic_segmentation_amd/preprocess.py:
 * resample_tables) and the kernels apply them in integer arithmetic, so the resized 8-bit image
 * equals Pillow's bit for bit; out = ((pixel / 255) - mean) / std in fp32 as torchvision computes it.
 *   src:     [H0][W0][3] uint8 RGB (device)         out: [3][H][W] fp32
 *   xcoef:   [W][xksize] int32, xbounds: [W][2] = {first input column, count}; NULL when W == W0
 *   ycoef:   [H][yksize] int32, ybounds: [H][2]; NULL when H == H0 (device pointers)
 *   mean, stdev: 3 floats each, HOST pointers (read during the call)
 *   workspace: mfl_preprocess_workspace_bytes() (the 8-bit image after the horizontal pass)
 */
size_t mfl_preprocess_workspace_bytes(int H0, int W0, int H, int W);
int mfl_preprocess_rgb8(const unsigned char *src, int H0, int W0, const int *xcoef, const int *xbounds, int xksize,
                        const int *ycoef, const int *ybounds, int yksize, const float *mean, const float *stdev,
                        float *out, int H, int W, void *workspace, size_t workspace_bytes, mfl_stream_t stream);

/*
 * Nearest-neighbour resize of an annotated image (8-bit label map), what DPNModel.fit applies to
 * the ground truth with PIL's default filter for palette images (AdvancedModel.py:265):
 * out[y][x] = src[yin[y]][xin[x]], the source indices computed on the host as Pillow does
 * (preprocess.py: nearest_indices).   src: [H0][W0], out: [H][W], yin: [H], xin: [W] int32 (device).
 */
int mfl_resize_labels(const unsigned char *src, int H0, int W0, const int *yin, const int *xin, unsigned char *out, int H,
                      int W, mfl_stream_t stream);

#ifdef __cplusplus
}
#endif

#endif /* MFL_H */
