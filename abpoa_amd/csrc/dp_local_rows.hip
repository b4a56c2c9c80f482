// Row-loop kernel for LOCAL alignment without a band (rows_local.h): amino-acid / short-read MSAs (BASELINE.json configs[4]).  One wavefront per
// alignment, every chunk of a row in registers; the chunk count is fixed per alignment (rows span the whole query), so the kernel picks one of
// three straight-line variants (3 / 5 / 9 chunks of 64 columns) once, outside the row loop.
#include <stdlib.h>
#include "engine_options.h"
#include "rows_local.h"

namespace abpoa_hip {

template <int GAP>
__global__ void __launch_bounds__(64) dp_local_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_local(b, d)) return;
    typedef int16_t T;
    const int lane = threadIdx.x & 63;
    FastIO<T> io;
    io.row_base = vgpr_ptr(b.row_base + d.row0); io.row_remain = vgpr_ptr(b.row_remain + d.row0);
    io.pred_off = vgpr_ptr(b.pred_off + d.poff0); io.pred_row = vgpr_ptr(b.pred_row + d.pred0);
    io.g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0); io.g_esn = vgpr_ptr(b.dp_end_sn + d.row0); io.row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    io.g_left = vgpr_ptr(b.left + d.row0); io.g_right = vgpr_ptr(b.right + d.row0); io.g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    io.planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = lane; i < d.qlen; i += 64) s_query[i] = g_query[i]; }
    __syncthreads();
    const int nch = ((d.qlen / 16 + 1) * 16 + 63) >> 6;
    if (nch <= 3) rows_local<T, GAP, 3>(b, d, io, s_query, b.out + a);
    else if (nch <= 5) rows_local<T, GAP, 5>(b, d, io, s_query, b.out + a);
    else rows_local<T, GAP, 9>(b, d, io, s_query, b.out + a);
}

// (Measured on configs[4], row loops per step: one wavefront 137 ms, teams of two 126 ms, of four 81 ms, of eight 128 ms -- 512 threads per workgroup halve the
//  residency.)
// Four wavefronts per alignment (rows_local_team): for launches that leave the GPU's SIMDs short of wavefronts -- BASELINE.json configs[4] is 1000 read-sets,
// one alignment each at a time: one wavefront per SIMD -- the chunks of a row are split over the four SIMDs of a CU.
constexpr int LOC_NW = 4;
template <int GAP>
__global__ void __launch_bounds__(64 * LOC_NW) dp_local_team_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_local(b, d)) return;
    typedef int16_t T;
    const int tid = threadIdx.x;
    FastIO<T> io;
    io.row_base = vgpr_ptr(b.row_base + d.row0); io.row_remain = vgpr_ptr(b.row_remain + d.row0);
    io.pred_off = vgpr_ptr(b.pred_off + d.poff0); io.pred_row = vgpr_ptr(b.pred_row + d.pred0);
    io.g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0); io.g_esn = vgpr_ptr(b.dp_end_sn + d.row0); io.row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    io.g_left = vgpr_ptr(b.left + d.row0); io.g_right = vgpr_ptr(b.right + d.row0); io.g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    io.planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = tid; i < d.qlen; i += 64 * LOC_NW) s_query[i] = g_query[i]; }
    __syncthreads();
    const int nch = ((d.qlen / 16 + 1) * 16 + 63) >> 6;
    if (nch <= LOC_NW) rows_local_team<T, GAP, 1, LOC_NW>(b, d, io, s_query, b.out + a);
    else if (nch <= 2 * LOC_NW) rows_local_team<T, GAP, 2, LOC_NW>(b, d, io, s_query, b.out + a);
    else rows_local_team<T, GAP, 3, LOC_NW>(b, d, io, s_query, b.out + a);
}

template <typename K>
static hipError_t launch_team(K kern, const DevBatch &b, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(b.n), dim3(64 * LOC_NW), (size_t)b.lds.total_local, stream, b);
    return hipGetLastError();
}

hipError_t launch_local_rows(const DevBatch &b, hipStream_t stream) {
    // teams while the launch has fewer alignments than the GPU has room for workgroups of four wavefronts at two per SIMD (8 wavefronts per CU x 256 CUs /
    // 4 = 2048 alignments would fill it; beyond ~1500 the single-wavefront kernel's SIMDs are busy anyway); ABPOA_HIP_LOCAL_TEAM=0 / 1 forces a form
    bool team = b.n <= 1536;
    { const char *e_ = opt_env("ABPOA_HIP_LOCAL_TEAM"); if (e_) team = atoi(e_) != 0; }
    if (team) return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_team(dp_local_team_kernel<1>, b, stream) : launch_team(dp_local_team_kernel<2>, b, stream);
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_one(dp_local_kernel<1>, b, stream, b.lds.total_local) : launch_one(dp_local_kernel<2>, b, stream, b.lds.total_local);
}

}  // namespace abpoa_hip
