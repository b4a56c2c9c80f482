// Wide-band row loop with TEAMS of wavefronts: NW = 2 or 4 wavefronts per alignment, each with a contiguous share of the chunks of every row
// (rows_fast.h, ilp_chunks with NW > 1: one LDS exchange + two barriers per row).  For launches with fewer alignments than SIMDs -- a single
// read-set, or the last few hundred of a job -- where one wavefront per alignment leaves most of the chip idle.
#include "rows_fast.h"

namespace abpoa_hip {

template <int GAP, int BITS, int NW>
__global__ void __launch_bounds__(NW * 64) dp_team_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || d.bits != BITS || !takes_wide(b, d)) return;
    align_fast_rows<typename std::conditional<BITS == 16, int16_t, int32_t>::type, GAP, NW, false>(b, d, b.out + a);
}

template <int GAP, int NW>
static hipError_t launch_team_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask & 1) e = launch_one(dp_team_kernel<GAP, 16, NW>, b, stream, b.lds.total_wide, NW * 64);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_team_kernel<GAP, 32, NW>, b, stream, b.lds.total_wide, NW * 64);
    return e;
}
hipError_t launch_team_rows(const DevBatch &b, hipStream_t stream) {
    const bool affine = b.gap_mode == ABPOA_HIP_AFFINE_GAP;
    if (b.lds.wide_nw == 2) return affine ? launch_team_gap<1, 2>(b, stream) : launch_team_gap<2, 2>(b, stream);
    if (b.lds.wide_nw == 4) return affine ? launch_team_gap<1, 4>(b, stream) : launch_team_gap<2, 4>(b, stream);
    return hipErrorInvalidValue;
}

}  // namespace abpoa_hip
