// Tail kernels of the fast path: global best + backtrack over the cell-record arenas the row loops left in HBM (backtrack.h).
#include "fast_tail.h"

namespace abpoa_hip {

template <int GAP, int BITS, bool DIR>
__global__ void __launch_bounds__(64) dp_fast_tail_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!(takes_fast(b, d) || takes_local(b, d)) || (BITS != 0 && d.bits != BITS)) return;
    if (BITS == 16 || (BITS == 0 && d.bits == 16)) align_fast_tail<int16_t, GAP, DIR>(b, d, b.out + a);      // (BITS == 0: both widths in one launch, see dp_wide_rows.hip)
    else align_fast_tail<int32_t, GAP, DIR>(b, d, b.out + a);
}

template <int GAP, bool DIR>
static hipError_t launch_tail_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask == 3) return launch_one(dp_fast_tail_kernel<GAP, 0, DIR>, b, stream, b.lds.total_tail);
    if (mask & 1) e = launch_one(dp_fast_tail_kernel<GAP, 16, DIR>, b, stream, b.lds.total_tail);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_tail_kernel<GAP, 32, DIR>, b, stream, b.lds.total_tail);
    return e;
}
hipError_t launch_fast_tail(const DevBatch &b, hipStream_t stream) {
    if (b.dir_mode) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1, true>(b, stream) : launch_tail_gap<2, true>(b, stream);
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1, false>(b, stream) : launch_tail_gap<2, false>(b, stream);
}

}  // namespace abpoa_hip
