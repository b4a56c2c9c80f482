// Row-loop kernels of the fast path (register-resident single-wave loop; see rows_fast.h), one kernel per gap mode and score width.
#include "rows_fast.h"

namespace abpoa_hip {

// The fast path is two kernels -- row loop, then global best + backtrack -- so that the row loop's register allocation
// (its SGPR budget above all) is not shared with the tail; the hand-over is the AlnOut record in HBM.
template <typename T, int GAP>
__device__ __forceinline__ void align_fast_rows(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
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
    long long cursor = 0, n_cells = 0; int status = 0, rows_done = 0, last_done = 0;
    const long long clk0 = (long long)__builtin_amdgcn_s_memtime();
    long long fseg[6] = {0, 0, 0, 0, 0, 0};
    rows_fast<T, GAP>(b, d, io, s_query, cursor, n_cells, status, rows_done, last_done, fseg);
    const long long clk1 = (long long)__builtin_amdgcn_s_memtime();
    if (lane == 0) { GLOBAL_AS AlnOut *o = vgpr_ptr(out_rec); o->status = status; o->n_cells = n_cells; o->cells_used = cursor; o->clk_dp = clk1 - clk0; o->n_rows_done = rows_done; for (int i_ = 0; i_ < 6; ++i_) o->seg[i_] = fseg[i_]; }
}

// One kernel per score width: the row loop of one width is ~40 KB of code, and a CU pair's 64 KB instruction cache has to hold
// what its 8 or so resident wavefronts execute; an alignment of the other width is left to the other kernel.
template <int GAP, int BITS>
__global__ void __launch_bounds__(64) dp_fast_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!takes_fast(b, d) || d.bits != BITS) return;           // dp_kernel's, or the other width's
    align_fast_rows<typename std::conditional<BITS == 16, int16_t, int32_t>::type, GAP>(b, d, b.out + a);
}

template <int GAP>
static hipError_t launch_rows_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask & 1) e = launch_one(dp_fast_kernel<GAP, 16>, b, stream, b.lds.total_rows);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_kernel<GAP, 32>, b, stream, b.lds.total_rows);
    return e;
}
hipError_t launch_fast_rows(const DevBatch &b, hipStream_t stream) {
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_rows_gap<1>(b, stream) : launch_rows_gap<2>(b, stream);
}

}  // namespace abpoa_hip
