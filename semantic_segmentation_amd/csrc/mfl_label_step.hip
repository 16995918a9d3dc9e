// Label transform + normalise step of the large-label mean-field path (33 .. 160 labels), fused, with the
// L x L label product on the matrix cores:
//   A    = aw * M + ab                       LinearActivation, AdvancedModel.py:52-56, 223
//   P    = compat . A                        the label-context product, AdvancedModel.py:143-144 with a 1x1 kernel
//   logQ = LogSoftmax(log_softmax(U) - P)    AdvancedModel.py:242-244
//   Q    = exp(logQ)  -> interior of the zero-padded planes the next gather reads
// At 150 labels the product is 22 500 multiply-adds per pixel against ~300 for the rest of the step: a dense
// [pixels x C] . [C x C] contraction.  It runs as v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: the
// 1e-4 bar rules out bf16 / xf32 inputs), which has the vector unit's peak rate but needs one instruction per
// 2048 multiply-adds instead of one per 64.
//
// Work split: a wave owns tiles of 32 consecutive pixels; lane (p, h) = (lane & 31, lane >> 5).
//   B operand (columns = pixels)          lane holds A[p][2j + h] for k-step j (read from M, coalesced)
//   A operand (rows = output channels)    lane holds compat[32 mb + p][2j + h] for channel block mb, from the
//                                         transposed zero-padded copy of the matrix in LDS (one per workgroup)
//   D                                     register v of block mb = P[p][32 mb + 8 (v / 4) + 4 h + v % 4]
// so the two lanes of a pixel hold interleaved quarters of its channels; the softmax maxima and sums are
// exchanged between them with one lane permute each.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/mfl.h"

namespace {

typedef float f16v __attribute__((ext_vector_type(16)));

// waves per workgroup (one workgroup per CU): the k-loop waits on its loads, so four waves per SIMD (128 registers,
// four of them spilled) beat two: 3.1 against 4.0 ms at 150 labels, 512 x 512, 16 images (12 waves: 3.4; loads of 16
// or 4 k-steps in flight instead of 8, or the next group's loads issued before this group's matrix instructions:
// no better)
#ifndef MFL_LS_WAVES
#define MFL_LS_WAVES 16
#endif
#ifndef MFL_LS_AHEAD
#define MFL_LS_AHEAD 8
#endif
constexpr int kWaves = MFL_LS_WAVES;

template <int MB>
__global__ void __launch_bounds__(64 * kWaves)
label_step_kernel(const float *__restrict__ msg, const float *__restrict__ aw, const float *__restrict__ ab,
                  const float *__restrict__ compat, const float *__restrict__ unary, float *__restrict__ logq,
                  float *__restrict__ q, int q_pad_k, int B, int C, int H, int W)
{
    extern __shared__ __attribute__((aligned(16))) float ct[];      // [2 * KJ][CP]: ct[c2][c] = compat[c][c2]
    constexpr int CP = 32 * MB;
    const int KJ = (C + 1) / 2;
    const int L = H * W;
    for (int i = threadIdx.x; i < 2 * KJ * CP; i += blockDim.x) {
        const int c2 = i / CP, c = i - c2 * CP;
        ct[i] = (c < C && c2 < C) ? compat[(size_t)c * C + c2] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int p = lane & 31, h = lane >> 5;
    const long long npix = (long long)B * L;
    const long long ntiles = (npix + 31) / 32;
    int Wp = W, p0 = 0;
    size_t qplane = (size_t)L;
    if (q_pad_k > 0) {
        Wp = W + q_pad_k - 1;
        p0 = q_pad_k / 2;
        qplane = (size_t)(H + q_pad_k - 1) * Wp;
    }
    auto other_half = [&](float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, v)));
    };
    // Every tensor is reached through a buffer descriptor over the whole tensor (the launcher checks that each
    // is smaller than 4 GB): a 32-bit per-lane offset (image, pixel, the lane's half) plus a uniform channel
    // offset in a scalar register; a lane offset past the tensor makes a load return 0 and drops a store.
    auto rsrc_of = [](const void *ptr, size_t bytes) {
        return __builtin_amdgcn_make_buffer_rsrc((void *)ptr, 0, (int)(uint32_t)bytes, 0x00020000);
    };
    const size_t ntot = (size_t)B * C * L * 4;
    const __amdgpu_buffer_rsrc_t rm = rsrc_of(msg, ntot), ru = rsrc_of(unary, ntot);
    const __amdgpu_buffer_rsrc_t rlq = rsrc_of(logq ? logq : msg, logq ? ntot : 0);
    const __amdgpu_buffer_rsrc_t rq = rsrc_of(q ? q : msg, q ? (size_t)B * C * qplane * 4 : 0);
    const __amdgpu_buffer_rsrc_t raw = rsrc_of(aw ? aw : msg, (size_t)C * L * 4), rab = rsrc_of(ab ? ab : msg, (size_t)C * L * 4);
    auto bload = [](__amdgpu_buffer_rsrc_t r, uint32_t voff, uint32_t soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)voff, (int)soff, 0));
    };
    const uint32_t L4 = (uint32_t)L * 4u, P4 = (uint32_t)qplane * 4u;
    const uint32_t oob = 0xfffffff0u;
    for (long long tile = (long long)blockIdx.x * kWaves + wave; tile < ntiles; tile += (long long)gridDim.x * kWaves) {
        const long long pix = tile * 32 + p;
        const bool ok = pix < npix;
        const long long pc = ok ? pix : npix - 1;
        const int b = (int)(pc / L), l = (int)(pc - (long long)b * L);
        const uint32_t vk = ((uint32_t)(b * C + h) * (uint32_t)L + (uint32_t)l) * 4u;       // channel h of the pixel (k-loop)
        const uint32_t vak = ((uint32_t)h * (uint32_t)L + (uint32_t)l) * 4u;                // ... in the [C][L] affine tensors
        f16v d[MB];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) d[mb][v] = 0.f;
        // The k-loop also streams the pixel's unary (channel 2j + h, like the message) into a running
        // log-sum-exp: its loads and exponentials hide behind the matrix instructions.
        float mrun = -INFINITY, srun = 0.f;
        // (k-steps in groups of kAhead: all loads of a group are issued before its matrix instructions, so that
        // a wave keeps 2 .. 4 x kAhead cache lines in flight instead of waiting for memory at every k-step)
        constexpr int kAhead = MFL_LS_AHEAD;
        for (int j0 = 0; j0 < KJ; j0 += kAhead) {
            float av[kAhead], uv[kAhead];
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                const int j = j0 + jj < KJ ? j0 + jj : KJ - 1;        // (past the end: the last k-step again, not used)
                const uint32_t so = (uint32_t)(2 * j) * L4;
                av[jj] = bload(rm, vk, so);
                uv[jj] = bload(ru, vk, so);
                if (aw) av[jj] = fmaf(bload(raw, vak, so), av[jj], bload(rab, vak, so));
            }
#pragma unroll
            for (int jj = 0; jj < kAhead; ++jj) {
                if (j0 + jj < KJ) {
                    const int c2 = 2 * (j0 + jj) + h;
                    const float a = c2 < C ? av[jj] : 0.f;
                    const float uu = c2 < C ? uv[jj] : -INFINITY;
                    const float *row = ct + c2 * CP + p;
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb) d[mb] = __builtin_amdgcn_mfma_f32_32x32x2f32(row[32 * mb], a, d[mb], 0, 0, 0);
                    const float mn = fmaxf(mrun, uu);
                    srun = srun * __expf(mrun - mn) + __expf(uu - mn);
                    mrun = mn;
                }
            }
        }
        float lse_u;
        {
            const float mo = other_half(mrun), so = other_half(srun);
            const float mm = fmaxf(mrun, mo);
            lse_u = mm + __logf(srun * __expf(mrun - mm) + so * __expf(mo - mm));
        }
        // the lane's output channels: block mb, register v -> c = 32 mb + 8 (v / 4) + 4 h + v % 4
        const uint32_t vo = ((uint32_t)(b * C + 4 * h) * (uint32_t)L + (uint32_t)l) * 4u;
        // z = (u - lse_u) - P, kept in the accumulator registers (channels past C: -inf)
        float mv = -INFINITY;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int c0 = 32 * mb + 8 * (v / 4) + (v % 4);
                const float uv = bload(ru, vo, (uint32_t)c0 * L4);
                const float z = c0 + 4 * h < C ? (uv - lse_u) - d[mb][v] : -INFINITY;
                d[mb][v] = z;
                mv = fmaxf(mv, z);
            }
        mv = fmaxf(mv, other_half(mv));
        float sv = 0.f;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) sv += __expf(d[mb][v] - mv);
        sv += other_half(sv);
        const float lse_v = mv + __logf(sv);
        const int y = l / W, x = l - y * W;
        const uint32_t vq = ((uint32_t)(b * C + 4 * h) * (uint32_t)qplane + (uint32_t)((y + p0) * Wp + x + p0)) * 4u;
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int v = 0; v < 16; ++v) {
                const int c0 = 32 * mb + 8 * (v / 4) + (v % 4);
                const bool valid = ok && c0 + 4 * h < C;
                const float lq = d[mb][v] - lse_v;
                if (logq) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, lq), rlq, (int)(valid ? vo : oob), (int)((uint32_t)c0 * L4), 0);
                if (q) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, __expf(lq)), rq, (int)(valid ? vq : oob), (int)((uint32_t)c0 * P4), 0);
            }
    }
}

template <int MB>
int launch(const float *msg, const float *aw, const float *ab, const float *compat, const float *unary, float *logq, float *q,
           int q_pad_k, int B, int C, int H, int W, hipStream_t st)
{
    const size_t lds = (size_t)2 * ((C + 1) / 2) * 32 * MB * 4;
    {
        // (the attribute belongs to the current device's copy of the kernel: set before every launch)
        const hipError_t e = hipFuncSetAttribute((const void *)label_step_kernel<MB>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return (int)e;
    }
    const long long ntiles = ((long long)B * H * W + 31) / 32;
    long long blocks = (ntiles + kWaves - 1) / kWaves;
    if (blocks > 256) blocks = 256;          // one workgroup per CU (its copy of the matrix takes most of the LDS); tiles are looped over
    hipLaunchKernelGGL(label_step_kernel<MB>, dim3((unsigned)blocks), dim3(64 * kWaves), lds, st, msg, aw, ab, compat, unary, logq, q,
                       q_pad_k, B, C, H, W);
    return (int)hipGetLastError();
}

}  // namespace

namespace mfl {

bool label_step_supported(int C) { return C > 32 && C <= 160; }

// (32-bit buffer offsets: every tensor below 4 GB)
bool label_step_fits(int B, int C, int H, int W, int q_pad_k)
{
    const unsigned long long plane = (unsigned long long)(H + (q_pad_k > 0 ? q_pad_k - 1 : 0)) * (W + (q_pad_k > 0 ? q_pad_k - 1 : 0));
    return (unsigned long long)B * C * plane * 4 < 0xfffffff0ull;
}

// msg: [B][C][H][W] window messages M; writes logq (optional) and q (optional; padded for a q_pad_k window or plain if 0).
int launch_label_step(const float *msg, const float *aw, const float *ab, const float *compat, const float *unary, float *logq,
                      float *q, int q_pad_k, int B, int C, int H, int W, hipStream_t st)
{
    if (!compat || !label_step_supported(C)) return MFL_ERR_INVALID_ARGUMENT;
    if (!label_step_fits(B, C, H, W, q_pad_k)) return MFL_ERR_TOO_LARGE;
    switch ((C + 31) / 32) {
    case 2: return launch<2>(msg, aw, ab, compat, unary, logq, q, q_pad_k, B, C, H, W, st);
    case 3: return launch<3>(msg, aw, ab, compat, unary, logq, q, q_pad_k, B, C, H, W, st);
    case 4: return launch<4>(msg, aw, ab, compat, unary, logq, q, q_pad_k, B, C, H, W, st);
    case 5: return launch<5>(msg, aw, ab, compat, unary, logq, q, q_pad_k, B, C, H, W, st);
    default: return MFL_ERR_INVALID_ARGUMENT;
    }
}

}  // namespace mfl
