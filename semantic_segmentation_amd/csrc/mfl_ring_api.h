// Internal interface between the C-ABI translation unit (mfl_api.hip) and the ring gather
// translation units (mfl_ring.hip, one per (CH, KS) pair so that they build in parallel).
#ifndef MFL_RING_API_H
#define MFL_RING_API_H

#include <hip/hip_runtime.h>

#include "mfl_ring.h"

namespace mfl {

struct RingArgs {
    RingGeom r;
    const float *xp;         // FS floats of slack, then padded planes [B][C][Hp][Wp], then slack
    const F4 *ip4;           // padded image [B][Hp][Wp]
    const uint32_t *table;   // ring_fill_record records
    const float *wtab;       // weight table (weights_kernel)
    float *wtab_out;         // mode 3: the table to write
    const float *pos_w, *col_w;
    const float *unary, *aw, *ab;             // mode 0
    const float *compat;                     // mode 0: zero-padded to (CH*NCS)^2, or NULL
    const float *compat_plain;               // mode 0: the caller's C x C matrix (tail pixels), or NULL
    float *logq, *qp_out;                    // mode 0: qp_out has the layout of xp
    float *y;                                // modes 1, 2
};

#define MFL_DECLARE_RING(CHV, KSV) int launch_ring_ch##CHV##_k##KSV(const RingArgs &a, int mode, hipStream_t st);
MFL_DECLARE_RING(8, 9)
MFL_DECLARE_RING(7, 11)
MFL_DECLARE_RING(10, 9)
MFL_DECLARE_RING(10, 11)

}  // namespace mfl

#endif  // MFL_RING_API_H
