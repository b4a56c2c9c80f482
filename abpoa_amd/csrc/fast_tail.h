// Tail of one alignment of the fast path: global best + backtrack over the cell-record arena its row loop left in HBM (backtrack.h);
// the hand-over from the row loop is the AlnOut record.  Used by the tail kernels (dp_fast_tail.hip) and by the all-rounds kernel (poa_rounds.hip).
#pragma once
#include "rows_fast.h"      // FastFmt (arena record format)
#include "rows_local.h"    // takes_local
#include "backtrack.h"
#include "backtrack_dir.h"

namespace abpoa_hip {

// role / spec_ctl / gen: two wavefronts on one direction-plane walk (backtrack_dir.h; the all-rounds kernel) -- -1: one wavefront, as everywhere else
template <typename T, int GAP, bool DIR = false>
__device__ __forceinline__ void align_fast_tail(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec, const int role = -1, int *spec_ctl = nullptr, const int gen = 0) {
    const int lane = threadIdx.x & 63;
    if (role >= 1) {      // helper: only where several wavefronts walk, and not before the main wavefront has finished the row loop and cleared the table
        if (!(DIR && takes_dir(b, d) && dir_walk_pair(b, d))) return;
        typedef __attribute__((address_space(3))) volatile int lds_vint_t;
        while (((lds_vint_t *)spec_ctl)[0] != gen) __builtin_amdgcn_s_sleep(4);
    }
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    const bool dir_a = DIR && takes_dir(b, d);      // (the direction-plane backtrack reads neither the score matrix nor -- but in its final pass, from HBM -- the query)
    if (!dir_a) { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < b.m * b.m; i += 64) s_mat[i] = g_mat[i]; }
    if (!dir_a) { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off); for (int i = lane; i < d.qlen; i += 64) s_query[i] = g_query[i]; }
    TailState ts;
    ts.status = out_rec->status; ts.n_cells = out_rec->n_cells; ts.cursor = out_rec->cells_used; ts.rows_done = out_rec->n_rows_done;
    ts.best_score = d.inf_min; ts.best_i = 0; ts.best_j = 0;
    if (b.align_mode == ABPOA_HIP_LOCAL_MODE || b.align_mode == ABPOA_HIP_EXTEND_MODE) { ts.best_score = out_rec->best_score; ts.best_i = out_rec->best_row; ts.best_j = out_rec->best_col; }      // (the local row loop and the extension-mode rows keep the best cell)
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = out_rec->seg[i_];
    ts.clk1 = (long long)__builtin_amdgcn_s_memtime(); ts.clk0 = ts.clk1 - out_rec->clk_dp;
    WG_SYNC();
    // (DIR: a launch in dir_mode -- its narrow-band alignments left direction words, its wide-band ones score records)
    if constexpr (DIR) {
        if (!takes_dir(b, d)) finish_alignment<T, GAP, FastFmt<T, GAP>::CW>(b, d, out_rec, ts);
        else if (role >= 0 && dir_walk_pair(b, d)) finish_alignment_dir<T, GAP, 64>(b, d, out_rec, ts, role, spec_ctl, gen);      // (several wavefronts on the walk: windows of 64 rows each)
        else if (role <= 0) finish_alignment_dir<T, GAP>(b, d, out_rec, ts);
    }
    else finish_alignment<T, GAP, FastFmt<T, GAP>::CW>(b, d, out_rec, ts);
}

}  // namespace abpoa_hip
