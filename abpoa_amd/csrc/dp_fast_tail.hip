// Tail kernels of the fast path: global best + backtrack over the cell-record arenas the row loops left in HBM (backtrack.h).
#include <algorithm>
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
    if (b.gap_mode == ABPOA_HIP_LINEAR_GAP) return launch_tail_gap<0, false>(b, stream);
    if (b.dir_mode == 2) {
        // every alignment of the launch walks direction words: no query, no score matrix in LDS -- the walk's image and its window start at byte 0 -- and the
        // window is what lets the whole launch be resident (LDS comes in pieces of 1280 B, 128 per CU; at most eight workgroups per CU: a walk is a chain
        // of dependent LDS reads, a second wavefront on the SIMD fills its waits)
        DevBatch t = b;
        const int per_cu = std::max(1, std::min(8, (b.n + 255) / 256));
        const int total = std::min((128 / per_cu) * 1280, t.lds.bt_off + 56 * 1024) & ~15;
        t.lds.phase_off = 0; t.lds.bt_bytes_tail = total - t.lds.bt_off; t.lds.total_tail = total;
        return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1, true>(t, stream) : launch_tail_gap<2, true>(t, stream);
    }
    if (b.dir_mode) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1, true>(b, stream) : launch_tail_gap<2, true>(b, stream);
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1, false>(b, stream) : launch_tail_gap<2, false>(b, stream);
}

}  // namespace abpoa_hip
