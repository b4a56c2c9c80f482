// Row-loop kernels for WIDE bands (10 kb reads: 240-300 columns = 4-5 chunks of 64 per row): every chunk of a row in registers at once
// (rows_fast.h, ilp_chunks).  This file: one wavefront per alignment; dp_team_rows.hip: teams of 2 / 4 wavefronts per alignment that share the
// chunks of every row (small batches: fewer alignments than SIMDs).  Rows the all-chunk body cannot take are done with the single-chunk bodies.
#include <stdio.h>
#include <stdlib.h>
#include "engine_options.h"
#include "rows_fast.h"

namespace abpoa_hip {

// (-DABPOA_HIP_WIDE_W3: the experiment of LOG.md round 5 -- three wavefronts per SIMD: 168 VGPRs, twelve workgroups per CU with a 2-row ring)
#ifdef ABPOA_HIP_WIDE_W3
#define WIDE_BOUNDS(NW) __launch_bounds__(NW * 64, 3)
#else
#define WIDE_BOUNDS(NW) __launch_bounds__(NW * 64)
#endif
template <int GAP, int BITS, int NW, bool DIR = false>
__global__ void WIDE_BOUNDS(NW) dp_wide_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || (BITS != 0 && d.bits != BITS) || !takes_wide(b, d)) return;
    // BITS == 0: both score widths in one launch.  A job whose graphs outgrow int16 on the way has a few rounds in which some read-sets are still
    // int16 and the others already int32; two launches (one per width) would run one after the other, each with the other's SIMDs idle.
    if (BITS == 16 || (BITS == 0 && d.bits == 16)) align_fast_rows<int16_t, GAP, NW, NW == 1, DIR>(b, d, b.out + a);
    else align_fast_rows<int32_t, GAP, NW, NW == 1, DIR>(b, d, b.out + a);
}

template <int GAP, int NW, bool DIR = false>
static hipError_t launch_wide_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    static bool told = false;
    if (!told && opt_env("ABPOA_HIP_VERBOSE")) {      // residency of the wide kernels on one CU
        told = true; int nb16 = 0, nb32 = 0;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb16, dp_wide_kernel<GAP, 16, NW>, NW * 64, (size_t)b.lds.total_wide);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb32, dp_wide_kernel<GAP, 32, NW>, NW * 64, (size_t)b.lds.total_wide);
        fprintf(stderr, "[abpoa-hip] wide row loop: %d wavefronts per alignment, %d B of LDS per workgroup, ring %d rows x %d " "columns, workgroups per CU: %d (int16) %d (int32)\n", NW,
                b.lds.total_wide, b.lds.wfr_rows, b.lds.wfr_cols, nb16, nb32);
    }
    if (mask == 3 && NW == 1) return launch_one(dp_wide_kernel<GAP, 0, NW, DIR>, b, stream, b.lds.total_wide, NW * 64);
    if (mask & 1) e = launch_one(dp_wide_kernel<GAP, 16, NW, DIR>, b, stream, b.lds.total_wide, NW * 64);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_wide_kernel<GAP, 32, NW, DIR>, b, stream, b.lds.total_wide, NW * 64);
    return e;
}
hipError_t launch_xl_rows(const DevBatch &b, hipStream_t stream);      // dp_xl_rows.hip
hipError_t launch_wide_rows(const DevBatch &b, hipStream_t stream) {
    if (b.lds.wide_nw == 1 && b.lds.wfr_cols == WIDE_RING_COLS_XL) return launch_xl_rows(b, stream);      // rows wider than 448 columns: the long-read form
    // (wide-band alignments keep their score records in dir_mode 1 and write direction words in dir_mode 2: dp_common.h takes_dir)
    if (b.lds.wide_nw == 1 && b.dir_mode == 2) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_wide_gap<1, 1, true>(b, stream) : launch_wide_gap<2, 1, true>(b, stream);
    if (b.lds.wide_nw == 1) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_wide_gap<1, 1>(b, stream) : launch_wide_gap<2, 1>(b, stream);
    return launch_team_rows(b, stream);      // dp_team_rows.hip (score-record arenas only: the host does not set dir_mode with teams)
}

}  // namespace abpoa_hip
