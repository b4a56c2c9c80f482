// Tail kernels of the fast path: global best + backtrack over the cell-record arenas the row loops left in HBM (backtrack.h).
#include "rows_fast.h"      // FastFmt (arena record format)
#include "rows_local.h"    // takes_local
#include "backtrack.h"

namespace abpoa_hip {

template <typename T, int GAP>
__device__ __forceinline__ void align_fast_tail(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    const int lane = threadIdx.x & 63;
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < b.m * b.m; i += 64) s_mat[i] = g_mat[i]; }
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = lane; i < d.qlen; i += 64) s_query[i] = g_query[i]; }
    TailState ts;
    ts.status = out_rec->status; ts.n_cells = out_rec->n_cells; ts.cursor = out_rec->cells_used; ts.rows_done = out_rec->n_rows_done;
    ts.best_score = d.inf_min; ts.best_i = 0; ts.best_j = 0;
    if (b.align_mode == ABPOA_HIP_LOCAL_MODE) { ts.best_score = out_rec->best_score; ts.best_i = out_rec->best_row; ts.best_j = out_rec->best_col; }      // (the local row loop keeps the best cell)
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = out_rec->seg[i_];
    ts.clk1 = (long long)__builtin_amdgcn_s_memtime(); ts.clk0 = ts.clk1 - out_rec->clk_dp;
    __syncthreads();
    finish_alignment<T, GAP, FastFmt<T, GAP>::CW>(b, d, out_rec, ts);
}

template <int GAP, int BITS>
__global__ void __launch_bounds__(64) dp_fast_tail_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (!(takes_fast(b, d) || takes_local(b, d)) || d.bits != BITS) return;
    align_fast_tail<typename std::conditional<BITS == 16, int16_t, int32_t>::type, GAP>(b, d, b.out + a);
}

template <int GAP>
static hipError_t launch_tail_gap(const DevBatch &b, hipStream_t stream) {
    const int mask = b.bits_mask ? b.bits_mask : 3;
    hipError_t e = hipSuccess;
    if (mask & 1) e = launch_one(dp_fast_tail_kernel<GAP, 16>, b, stream, b.lds.total_tail);
    if (e == hipSuccess && (mask & 2)) e = launch_one(dp_fast_tail_kernel<GAP, 32>, b, stream, b.lds.total_tail);
    return e;
}
hipError_t launch_fast_tail(const DevBatch &b, hipStream_t stream) {
    return b.gap_mode == ABPOA_HIP_AFFINE_GAP ? launch_tail_gap<1>(b, stream) : launch_tail_gap<2>(b, stream);
}

}  // namespace abpoa_hip
