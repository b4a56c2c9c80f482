// Row-loop kernels of the fast path (register-resident single-wave loop; see rows_fast.h), one kernel per gap mode and score width.
#include "rows_fast.h"

namespace abpoa_hip {

// One kernel per score width: the row loop of one width is ~40 KB of code, and a CU pair's 64 KB instruction cache has to hold
// what its 8 or so resident wavefronts execute; an alignment of the other width is left to the other kernel.
// DIR: direction-plane arenas (rows_fast.h DirFmt) instead of score records; the launch picks the instantiation by DevBatch.dir_mode
template <int GAP, int BITS, bool DIR>
__global__ void __launch_bounds__(64) dp_fast_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || (BITS != 0 && d.bits != BITS) || takes_wide(b, d)) return;           // dp_kernel's, the other width's, or the wide row loop's
    if (BITS == 16 || (BITS == 0 && d.bits == 16)) align_fast_rows<int16_t, GAP, 1, false, DIR>(b, d, b.out + a);      // (BITS == 0: both widths in one launch, see dp_wide_rows.hip)
    else align_fast_rows<int32_t, GAP, 1, false, DIR>(b, d, b.out + a);
}

template <int GAP, bool DIR>
static hipError_t launch_rows_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask == 3) return launch_one(dp_fast_kernel<GAP, 0, DIR>, b, stream, b.lds.total_rows);
    if (mask & 1) e = launch_one(dp_fast_kernel<GAP, 16, DIR>, b, stream, b.lds.total_rows);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_kernel<GAP, 32, DIR>, b, stream, b.lds.total_rows);
    return e;
}
hipError_t launch_fast_rows(const DevBatch &b, hipStream_t stream) {
    if (b.gap_mode == ABPOA_HIP_LINEAR_GAP) return launch_rows_gap<0, false>(b, stream);      // (H records only: no direction words)
    if (b.dir_mode) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_rows_gap<1, true>(b, stream) : launch_rows_gap<2, true>(b, stream);
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_rows_gap<1, false>(b, stream) : launch_rows_gap<2, false>(b, stream);
}

}  // namespace abpoa_hip
