// Internal interface between the C-ABI translation unit (mfl_api.hip) and the diagonal gather
// translation units (mfl_diag.hip, one per (CH, KS) pair so that they build in parallel).
#ifndef MFL_DIAG_API_H
#define MFL_DIAG_API_H

#include <hip/hip_runtime.h>

#include "mfl_diag.h"

namespace mfl {

struct DiagArgs {
    DiagGeom g;
    int epi_rows;            // b rows of the workgroup tile that fit in LDS per epilogue pass
    size_t lds_bytes;        // dynamic LDS of the launch
    const float *xp;         // FS floats of slack, then padded planes [B][C][Hp][Wp], then slack
    const F4 *ip4;           // padded image [B][Hp][Wp]
    const uint32_t *table;   // diag_fill_record records
    const uint32_t *groups;  // diag_groups list
    const float *pos_w, *col_w;
    const float *unary, *aw, *ab;             // mode 0
    const float *compat;                     // mode 0: zero-padded to (CH*NCS)^2, or NULL
    const float *compat_plain;               // mode 0: the caller's C x C matrix (tail pixels), or NULL
    float *logq, *qp_out;                    // mode 0: qp_out has the layout of xp
    float *y;                                // mode 1
    size_t copy_stride;                      // mode 3: qp_out holds 2k copies of the gradient planes, this many floats apart
    unsigned long long *dbg;                 // cycle stamps of diagnostic builds (-DMFL_STAMPS), else NULL
};


#define MFL_DECLARE_DIAG(CHV, KSV) int launch_diag_ch##CHV##_k##KSV(const DiagArgs &a, int mode, hipStream_t st);
#define MFL_DECLARE_DIAG_K(KSV) MFL_DECLARE_DIAG(8, KSV) MFL_DECLARE_DIAG(10, KSV) MFL_DECLARE_DIAG(11, KSV)
MFL_DECLARE_DIAG_K(3)
MFL_DECLARE_DIAG_K(5)
MFL_DECLARE_DIAG_K(7)
MFL_DECLARE_DIAG(5, 9)
MFL_DECLARE_DIAG(5, 11)

}  // namespace mfl

#endif  // MFL_DIAG_API_H
