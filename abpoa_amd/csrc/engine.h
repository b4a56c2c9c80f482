// Internal interface between the host engine (engine.cpp) and the device code (dp_kernel.hip).
// Not installed; the public C-ABI is include/abpoa_hip.h.
#pragma once
#include <stdint.h>
#include <hip/hip_runtime.h>
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

// Per-alignment descriptor, resident in HBM for the duration of a launch.
struct AlnDesc {
    int32_t n_rows, qlen;
    int32_t bits;        // 16 | 32  (reference: simd_abpoa_align.c:1672-1683)
    int32_t inf_min;
    int32_t w;           // band half-width, reference :445
    int32_t cigar_cap;   // entries
    int32_t flags;       // ALN_FAST_OK: every row active and max_pos_left/right start as (n_rows, 0) -> the register-resident row loop may be used
    int32_t pad0;        // expected extra band columns (device-resident driver, PoaSet.band_extra; 0 elsewhere): dp_common.h takes_wide
    int64_t query_off;   // into query pool (bytes)
    int64_t row0;        // index of DP row 0 in every per-row pool
    int64_t poff0;       // index of pred_off[0] / out_off[0] in the (n_rows+1)-sized offset pools
    int64_t pred0;       // index of this alignment's first entry in pred_row pool
    int64_t out0;        // same for out_row pool
    int64_t plane_off;   // BYTE offset of this alignment's score-plane arena
    int64_t plane_cap;   // arena capacity in cells (of `bits` width)
    int64_t cigar_off;   // index into cigar pool (uint64 words)
};

// Per-alignment result record.
struct AlnOut {
    int32_t status;
    int32_t best_score, best_row, best_col;
    int32_t node_s, node_e, query_s, query_e;
    int32_t n_aln_bases, n_matched_bases;
    int32_t n_cigar;
    int32_t pad;
    int64_t n_cells;      // sum (end_sn-beg_sn+1)*pn over rows 1..gn-2
    int64_t cells_used;   // arena cells consumed (all rows incl. row 0, all planes)
    int64_t clk_dp, clk_bt; // shader-clock ticks spent in the row loop / in the backtrack (s_memtime)
    int32_t n_rows_done, n_bt_steps;
    int64_t seg[6];       // ABPOA_HIP_PROFILE builds: ticks per row-loop segment
};

#define ALN_FAST_OK 1
#define ALN_SKIP 2       // nothing to align (device-resident driver: a set that is done, or fell back): every kernel leaves the descriptor alone
#define WIDE_RING_COLS 448   // score-ring columns of the single-wave wide kernel (7 chunks of 64)
#define WIDE_RING_COLS_XL 704   // ... of its long-read form (11 chunks: band half-widths up to 343 = reads up to the fast loops' 32 000 bases)
#define ABPOA_HIP_STATUS_OVERFLOW 1   // arena too small: host retries with a full-width arena
#define ABPOA_HIP_STATUS_NEED_SCORES 2   // direction-plane arenas (dir_plane.h): the backtrack met the one case the plane cannot decide -> redo with score records

// LDS carve-up of one wavefront (= one workgroup), chosen on the host per launch.  Byte offsets from the
// dynamic-LDS base; every region is 16-byte aligned.
struct LdsPlan {
    int32_t q_off, q_cap;         // query codes (uint8), used when qlen <= q_cap
    int32_t mat_off;              // int32 [m*m]
    int32_t phase_off;            // start of the region shared by the DP phase and the backtrack phase
    // --- DP phase (offsets from phase_off) ---
    int32_t ring_rows, ring_cols; // recent-row score ring: [ring_rows][planes_in_ring][ring_cols] cells of 4 bytes max
    int32_t ring_off;             // (relative to phase_off)
    // --- backtrack phase ---
    int32_t bt_off, bt_bytes;     // arena tile (relative to phase_off)
    int32_t bt_bytes_tail;        // arena window of the fast path's tail kernel (bt_bytes is the general kernel's, which shares the region with its score ring)
    int32_t bt_wc;                // column-slice windows of the backtrack: columns per slice (0: the kernel's default)
    // --- fast row loop (dp_kernel.hip rows_fast): packed H|E score ring [fr_rows][words][fr_cols + 4] dwords at phase_off + fr_off
    int32_t fr_off, fr_rows, fr_cols;   // fr_rows: power of two <= 64; fr_cols: multiple of 64, 0 = fast loop disabled
    // --- wide row loop (NW wavefronts per alignment, rows_fast<.., NW>): its own score ring [wfr_rows][words][wfr_cols + 4] at phase_off + fr_off,
    //     then the exchange slots (wx_off, relative to phase_off); wide_nw = 0: not used by this launch
    int32_t wfr_rows, wfr_cols, wx_off, wide_nw;
    int32_t wide_w_lo, wide_w_hi; // an alignment takes the wide loop iff wide_w_lo <= its band half-width w <= wide_w_hi
    int32_t narrow_off;           // 1: every alignment of the launch takes the wide loop -- the narrow row-loop kernels are not launched
    int32_t total_wide;           // dynamic LDS bytes of the wide row-loop kernel
    // (the wide kernels' own carve-up: query at q_off packed two 4-bit codes to a byte, then the extended matrix at w_mx_off, then -- w_phase_off -- ring and exchange slots)
    int32_t w_mx_off, w_phase_off;
    // --- local row loop (rows_local.h): ring [loc_rows][words][loc_cols + 4] at phase_off + fr_off; loc_cols = 0: not used by this launch
    int32_t loc_rows, loc_cols, total_local;
    int32_t mx_off;               // int32 [m*(m+1)]: score matrix with an extra all-zero query column (code m = "no query base")
    int32_t total;                // dynamic LDS bytes to request (general kernel: union of every phase)
    int32_t total_rows, total_tail;   // the two fast-path kernels request only what their phase needs (4+ workgroups per CU must fit)
};

// Everything one launch needs; passed by value as the kernel argument.
struct DevBatch {
    int32_t n;
    int32_t m;
    int32_t o1, e1, o2, e2;
    int32_t align_mode, gap_mode, wb, zdrop, ret_cigar, rev_cigar;
    int32_t want_trace;          // also record the per-row arg-max column (tests)
    int32_t dbg;                 // ablation switches for timing experiments (env ABPOA_HIP_DBG); 0 in production
    int32_t fresh_band;          // max_pos_left/right start as (n_rows, 0): initialise them on the device
    int32_t want_lr;             // the caller reads max_pos_left/right back (the fast row loop derives them in a post-pass)
    int32_t bits_mask;           // score widths that may occur among the fast alignments: 1 = int16, 2 = int32, 3 = both (one kernel per width)
    int32_t dir_mode;            // 1: the narrow-band fast row loops write direction words (dir_plane.h) instead of score records, the tail walks those; 2: the wide-band ones too
    int32_t pad1;
    LdsPlan lds;
    const int32_t *mat;          // [m*m]
    const AlnDesc *aln;          // [n]
    AlnOut *out;                 // [n]
    const uint8_t *query;        // pool
    const uint8_t *row_base;     // per-row pools ...
    const int32_t *row_node_id;
    const int32_t *row_remain;
    const uint8_t *row_active;
    const uint8_t *row_sdist;    // dir_mode: per row min(255, largest row distance to a successor), 255 for a predecessor of the sink: rows whose H / E
                                 // some later row (beyond the LDS score ring) or the global best will read from HBM keep a score record besides the direction words
    const uint32_t *row_pd;      // dir_mode: per row TWO dwords: the row distances to its first eight predecessors, a byte each (255: none / further than 254 rows):
                                 // what a step of the direction-plane backtrack needs to find the row it moves to (backtrack_dir.h); later ones come from the CSR arrays
    const int32_t *pred_off;     // (n_rows+1) per alignment
    const int32_t *pred_row;
    const int32_t *out_off;
    const int32_t *out_row;
    int32_t *left, *right;       // in/out
    int32_t *dp_beg_sn, *dp_end_sn;   // out, per row (-1 = never computed)
    int64_t *row_cell_off;       // out, per row: first cell of the row inside the alignment's arena
    int32_t *row_max_i;          // out, per row
    uint8_t *planes;             // arena pool (bytes)
    uint64_t *cigar;             // pool
};

// Which jobs / alignments the banded global row loops (rows_fast.h) take -- shared by the kernels (dp_common.h takes_fast) and their host mirrors.
// (linear gaps, round 5: gap extension >= 1 -- the row arg-max is taken before the in-row scan -- and band half-widths of the narrow loop only: LINEAR_FAST_W; the
//  wide kernels have no linear form; local / band-less linear jobs stay with the general kernel)
constexpr int LINEAR_FAST_W = 40;      // = LdsPlan.wide_w_lo's default
__host__ __device__ inline bool fast_global_job(int gap_mode, int align_mode, int wb, int e1) {
    if (wb < 0) return false;
    if (gap_mode == ABPOA_HIP_LINEAR_GAP && e1 < 1) return false;
    return align_mode == ABPOA_HIP_GLOBAL_MODE || align_mode == ABPOA_HIP_EXTEND_MODE;
}
__host__ __device__ inline bool fast_global_aln(int gap_mode, int w, int pad0) { return gap_mode != ABPOA_HIP_LINEAR_GAP || w + (pad0 >> 1) < LINEAR_FAST_W; }

// Fixed LDS structures of the kernel (rows per metadata tile, ring depths); see dp_kernel.hip.
int lds_fixed_bytes_dp();
int lds_fixed_bytes_bt();

// Launches the DP kernel for the whole batch on `stream`.
hipError_t launch_dp(const DevBatch &b, int n_fast, hipStream_t stream, hipEvent_t after_rows);
hipError_t launch_dp_fast(const DevBatch &b, hipStream_t stream, hipEvent_t after_rows);
hipError_t launch_dp_general(const DevBatch &b, hipStream_t stream);      // the general kernel alone (device-resident driver, jobs outside the fast row loops)

}  // namespace abpoa_hip
