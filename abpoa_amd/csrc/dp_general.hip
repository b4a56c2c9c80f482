// General kernel (any mode, any gap model, band on or off; plane-major arenas) and the launch entry points of the engine.
#include "rows_general.h"
#include "rows_local.h"
#include "backtrack_dir.h"

namespace abpoa_hip {

int lds_fixed_bytes_dp() { return (int)((sizeof(DpLds) + 15) & ~15u); }
int lds_fixed_bytes_bt() { return (int)(((sizeof(BtLds) > sizeof(DirBt) ? sizeof(BtLds) : sizeof(DirBt)) + 15) & ~15u); }      // (the two backtracks' row / edge tables; the arena window follows)

template <int GAP>
__global__ void __launch_bounds__(64) dp_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if ((d.flags & ALN_SKIP) || takes_fast(b, d) || takes_local(b, d)) return;            // nothing to do, or the fast row loops'
    if (d.bits == 16) align_one<int16_t, GAP>(b, d, b.out + a);
    else align_one<int32_t, GAP>(b, d, b.out + a);
}

hipError_t launch_general(const DevBatch &b, hipStream_t stream) {
    switch (b.gap_mode) {
        case ABPOA_HIP_LINEAR_GAP: return launch_one(dp_kernel<0>, b, stream, -1);
        case ABPOA_HIP_AFFINE_GAP: return launch_one(dp_kernel<1>, b, stream, -1);
        default: return launch_one(dp_kernel<2>, b, stream, -1);
    }
}

hipError_t launch_dp_general(const DevBatch &b, hipStream_t stream) { return b.n <= 0 ? hipSuccess : launch_general(b, stream); }

// The fast path is two kernels -- row loop, then global best + backtrack -- so that the row loop's register allocation
// (its SGPR budget above all) is not shared with the tail; the hand-over is the AlnOut record in HBM.
hipError_t launch_dp_fast(const DevBatch &b, hipStream_t stream, hipEvent_t after_rows) {
    if (b.n <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (b.align_mode != ABPOA_HIP_LOCAL_MODE && !b.lds.narrow_off) e = launch_fast_rows(b, stream);
    if (e == hipSuccess && b.lds.wide_nw >= 1 && b.gap_mode != ABPOA_HIP_LINEAR_GAP) e = launch_wide_rows(b, stream);
    if (e == hipSuccess && b.lds.loc_cols > 0 && b.align_mode == ABPOA_HIP_LOCAL_MODE) e = launch_local_rows(b, stream);
    if (e == hipSuccess) e = hipEventRecord(after_rows, stream);
    if (e == hipSuccess) e = launch_fast_tail(b, stream);
    return e;
}

// n_fast: how many alignments of the batch take the fast row loop (engine.cpp applies takes_fast() on the host); a kernel
// with nothing to do is not launched.
hipError_t launch_dp(const DevBatch &b, int n_fast, hipStream_t stream, hipEvent_t after_rows) {
    if (b.n <= 0) return hipSuccess;
    hipError_t e = hipSuccess;
    if (n_fast > 0) e = launch_dp_fast(b, stream, after_rows);
    if (e == hipSuccess && n_fast < b.n) e = launch_general(b, stream);
    return e;
}

}  // namespace abpoa_hip
