// Internal interface between the C-ABI translation unit (mfl_api.hip) and the
// gather-kernel translation units (mfl_gather.hip, compiled once per channel-chunk
// width CC with -DMFL_CC=<width> so that the widths build in parallel).
#ifndef MFL_GATHER_H
#define MFL_GATHER_H

#include <hip/hip_runtime.h>

#include "mfl_core.h"

namespace mfl {

constexpr int kThreads = 256;

struct GatherArgs {
    Geom g;                 // g.M selects the pixels-per-thread variant
    bool from_cd;           // weights from a colour-distance tensor instead of the padded image
    bool tab_lds;           // offset table staged in LDS (else read from global memory)
    const float *xp;        // padded planes [B][C][Hp][Wp]
    const F4 *ip4;          // padded image  [B][Hp][Wp]      (from_cd == false)
    const float *cd;        // [B][L][K]                       (from_cd == true)
    const int *tab;         // CK offsets
    const float *pos_w, *col_w;
};

// The shapes the tap-split forward kernel takes (local_forward_taps_kernel): a window of at least 16 x 16 taps and so
// few pixels that one pixel per thread cannot fill the GPU (the reference model's own 64 x 64 image with a 50 x 50
// window).  The mean-field loop then runs unfused on these shapes: window products, label transform, combine.
inline bool taps_kernel_applies(const Geom &g)
{
#ifdef MFL_NO_TAPS_KERNEL
    return false;       // comparison builds
#endif
    return g.K >= 256 && (long long)g.B * g.L <= 64 * 1024;
}

struct BackwardArgs {
    const float *dy;
    float *dxd, *dxp, *partials, *dcd;
};

// tap-split backward (local_backward_taps_kernel): plain window sums P, Q and the fixed-point gradient planes
struct TapsBackwardArgs {
    const float *dy;
    const float *scale;             // the fixed-point scale (a power of two), device pointer
    float *P, *Q;                   // [B][C][L]
    unsigned long long *dxp;        // zeroed padded planes [B][C][Hp][Wp] of 64-bit sums
};

struct StepArgs {
    const float *unary, *aw, *ab, *compat;
    float *logq, *qp_out;
};

#define MFL_DECLARE_GATHER(CCV)                                                                            \
    int launch_local_forward_cc##CCV(const GatherArgs &a, float *y, hipStream_t st);                       \
    int launch_local_backward_cc##CCV(const GatherArgs &a, const BackwardArgs &b, hipStream_t st);         \
    int launch_local_backward_taps_cc##CCV(const GatherArgs &a, const TapsBackwardArgs &t, hipStream_t st); \
    int launch_meanfield_step_cc##CCV(const GatherArgs &a, const StepArgs &s, hipStream_t st);

MFL_DECLARE_GATHER(8)
MFL_DECLARE_GATHER(19)
MFL_DECLARE_GATHER(21)
MFL_DECLARE_GATHER(32)

}  // namespace mfl

#endif  // MFL_GATHER_H
