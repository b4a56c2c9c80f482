#pragma once
#include "dp_common.h"
#include "dir_plane.h"
#include "rows_fast.h"      // DirFmt, FastFmt
#include "backtrack.h"      // TailState

namespace abpoa_hip {

// Global best + backtrack over DIRECTION-PLANE arenas (dir_plane.h): the row loops left one word per cell that records every comparison the
// reference backtrack (src/simd_abpoa_align.c:109-429) would make there, so the walk never reads a score.  oracle/dir_model.c is the CPU statement of
// exactly this walk (checked against the value-comparing backtrack on every golden).
//
// Window: rows [lo, hi] of words staged in LDS by LDS-DMA, up to DBTR rows, and per row
//     rec  { p0, p1 }: for each of the first two predecessors its row distance (a byte; 255 = none / further than 254 rows / not in the window) and,
//          above it, 16 bits, half the byte offset AB of its words in the window: the word of column c sits at window + AB + (c - cref) * DB;
//     pd   the row distances to the first eight predecessors, a byte each (DevBatch.row_pd, made once per alignment by the graph phase);
//     stg  first staged column | left-cut flag << 15 | staged columns << 16;   rowA  the row's own AB
//   so a match step -- 85-97 % of a walk -- needs ONE LDS round trip (the cell's word and its row's rec, both at addresses the previous step already
//   knew) and a dozen scalar instructions: the word names the predecessor (kM), rec gives its row distance and where its words are.  Any other
//   step, and a match through the third or a later predecessor, takes the full step below (pd / stg / rowA).  No cross-lane traffic anywhere;
//   node ids, the matched-base count and the reversal are one lane-parallel pass over the finished cigar.
//   * narrow bands: whole rows, ONE contiguous copy (rows are adjacent in the arena; a row that also keeps its score records drags them along): 256 rows
//     of 1 kb reads are 32 KB;
//   * wide bands: a triangle of column slices, one DMA per row -- a match step goes one column back and at least one row up, a deletion only up, so from
//     (hi, jtop) the walk reaches columns [jtop - (hi - r), jtop] of row r unless insertions push it further left: row r stages DIR_TRI_SLACK more
//     columns than that (the rows-per-column ratio of a path, 1.2 - 2.4, adds slack of its own).  A walk that leaves a slice re-centres the window.
//   Runs of match steps record only the ROW they pass (one v_writelane); their cigar words are made 64 at a time, a lane each.
// The one situation the plane cannot decide (dir_plane.h: F origin under an F-term H with dF > o) ends the walk with ABPOA_HIP_STATUS_NEED_SCORES and
// the host redoes that alignment with score records; oracle/dir_model.c counts it: 0 in 250 000 steps of noisy 1-5 kb reads.
constexpr int DBTR = 256;     // most rows of a window
constexpr int DIR_TRI_SLACK = 9;
constexpr int DIR_NA = -32768;
template <int R> struct __attribute__((aligned(16))) DirBtT { int2 rec[R]; int2 pd[R]; int32_t stg[R], rowA[R]; };      // the walk's LDS image (R rows at most in a window); the words follow
typedef DirBtT<DBTR> DirBt;

// Two wavefronts on one walk (the all-rounds kernel, whose workgroups have three wavefronts idle during the backtrack): the MAIN wavefront (role 0) walks
// from the best cell as always; a HELPER (role 1) starts at the same time from the arg-max cell of a row in the middle of the graph, in the state every
// match step leaves behind (all operations allowed, indel_first 0), and walks to the end.  Backtracks that pass through the same cell in the same state
// are identical from there on, and paths from neighbouring cells of a row run into each other within a few rows; both wavefronts note the cells of their
// match runs in a table in LDS (row -> column | index of the cigar word), and when the main walk steps on a cell the helper has been on it stops:
// the rest of its cigar is the helper's, from that word on.  No merge (the helper started on a branch the real path never touches): the main
// wavefront simply walks to the end itself.  spec_ctl: eight ints of static LDS (hand-over flags and the helper's results); gen: the round, != 0.
constexpr int SPEC_PM_ROWS = 256;      // rows below a helper's start row that its table covers
constexpr int SPEC_WK = 4;             // wavefronts on a walk: the main one and three helpers, which start at 3/4, 1/2 and 1/4 of the graph
// (what both wavefronts can tell before the row loop has finished: graphs of at least 768 rows; cigar indices and columns must fit the table's 16-bit fields)
__device__ __forceinline__ bool dir_walk_pair(const DevBatch &b, const AlnDesc &d) { return d.n_rows >= 768 && d.cigar_cap < 65536 && d.qlen < 65536 && b.ret_cigar; }
// More than one helper: helper r notes its cells in table r and looks the cells of its own runs up in the table of the next wavefront down whose rows it has
// reached, exactly as the main wavefront does; the cigar is then a chain -- main, then from the word where main met helper a on, then from where a met b ...
// spec_ctl: 8 ints per wavefront (ints 0..7: [0] = "tables are clear", set by the main wavefront; helper r: [8r] done, status, words, j, start_i, start_j,
// steps, met << 16 | index -- -1: walked to the end).  DBR: rows a window holds at most (64 per wavefront when four share the backtrack's LDS).
template <typename T, int GAP, int DBR = DBTR>
__device__ __forceinline__ void finish_alignment_dir(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec, const TailState &ts, const int role = -1,
                                                     int *spec_ctl = nullptr, const int gen = 0) {
    constexpr int PN = Width<T>::PN, CW = FastFmt<T, GAP>::CW, DB = DirFmt<T, GAP>::DB, S = (int)sizeof(T);
    constexpr int ALIGN = 16 / DB;                    // columns per 16-byte piece of a row of words
    constexpr int DBL = DB == 2 ? 1 : 2;
    constexpr int NQ = DBR / 64;
    typedef DirBtT<DBR> DirBt;
    typedef typename std::conditional<GAP == 1, uint16_t, uint32_t>::type DW;
    typedef __attribute__((address_space(3))) unsigned char lds_byte_t;
    typedef __attribute__((address_space(3))) DW lds_dw_t;
    typedef int v2i_t __attribute__((ext_vector_type(2)));
    typedef __attribute__((address_space(3))) v2i_t lds_i2_t;
    typedef __attribute__((address_space(3))) int lds_i32_t;
    const int lane = threadIdx.x & 63;
    const int gn = d.n_rows, qlen = d.qlen;
    const int o1 = b.o1, o2 = b.o2;
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const uint32_t *row_pd = vgpr_ptr(b.row_pd + 2 * d.row0);      // (two dwords per row)
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);
    const unsigned char *arena = (const unsigned char *)planes;
    int status = ts.status, best_score = ts.best_score, best_i = ts.best_i, best_j = ts.best_j, bt_steps = 0;
    WG_SYNC();       // all of this wave's arena / band stores have landed before the loads below
    // ---- two wavefronts on the walk?
    typedef __attribute__((address_space(3))) volatile int lds_vint_t;
    const bool spec_static = role >= 0 && dir_walk_pair(b, d), spec = spec_static && status == 0;
    // helper r starts at row gn (4 - r) / 4; its table covers the 256 rows below
    auto start_row = [&](int r_) __attribute__((always_inline)) { return (int)(((long long)gn * (SPEC_WK - r_)) / SPEC_WK); };
    const int R_g = spec && role >= 1 ? start_row(role) : 0, R_lo = R_g - SPEC_PM_ROWS + 1;
    const int total_lds = b.lds.bt_off + b.lds.bt_bytes_tail;
    const int half_lds = spec_static ? ((total_lds - (SPEC_WK - 1) * SPEC_PM_ROWS * 4) / SPEC_WK) & ~15 : total_lds;      // each walk's share of the backtrack region; the tables sit behind them
    int *pm_all = (int *)(lds_raw + b.lds.phase_off + SPEC_WK * half_lds);      // table of helper r: pm_all + (r - 1) * SPEC_PM_ROWS
    int *pm = pm_all + (role >= 1 ? role - 1 : 0) * SPEC_PM_ROWS;
    lds_vint_t *ctl = (lds_vint_t *)spec_ctl, *my = ctl + 8 * (role > 0 ? role : 0);
    auto helper_out = [&](int st_) __attribute__((always_inline)) { if (lane == 0) my[1] = st_; asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); if (lane == 0) my[0] = gen; };
    if (role >= 1) {
        // helper (it has waited for the main wavefront's "tables are clear", which also says the row loop's stores have landed: fast_tail.h): the start cell
        if (!spec) { if (spec_static) helper_out(-1); return; }
        const int c_g = __builtin_amdgcn_readfirstlane(gld_i32(vgpr_ptr(b.row_max_i + d.row0) + R_g));
        if (c_g < 1 || c_g > qlen) { helper_out(-1); return; }      // (no usable start: status -1 = "no helper")
        best_i = R_g; best_j = c_g;
    }

    // ------------------------------------------------------------------ global best, reference :1028-1041 (the sink's predecessors keep their score records)
    // (records of the wide kernel's spill rows are compact: rows_fast.h CWR)
    const int cwr = takes_wide(b, d) ? (sizeof(T) == 2 ? (GAP == 2 ? 4 : 2) : 2) : CW;
    if (status == 0 && role < 1) {
        // a lane per in-edge of the sink, 64 at a time (the three dependent loads of an edge are in flight for all of them together); the first maximum in
        // list order wins, as in the reference's loop with its strict ">"
        const int k0 = __builtin_amdgcn_readfirstlane(pred_off[gn - 1]), k1 = __builtin_amdgcn_readfirstlane(pred_off[gn]);
        for (int kb = k0; kb < k1; kb += 64) {
            const int k = kb + lane; const bool v = k < k1;
            int in_row = 0, end = 0, score = INT_MIN;
            if (v) {
                in_row = pred_row[k];
                const int pe = g_esn[in_row], pb = g_bsn[in_row];
                const int dpe = (pe + 1) * PN - 1; end = qlen > dpe ? dpe : qlen;
                score = (int)planes[g_coff[in_row] - (long long)(pe - pb + 1) * cwr * PN + (long long)(end - pb * PN) * cwr];      // (its records sit in front of its words)
            }
            const int mx = wave_max_i32(score);
            if (mx > best_score) {
                const int first = __builtin_ctzll(__ballot(v && score == mx));
                best_score = mx; best_i = __builtin_amdgcn_readlane(in_row, first); best_j = __builtin_amdgcn_readlane(end, first);
            }
        }
    }

    int n_cigar = 0, node_s = 0, node_e = 0, query_s = 0, query_e = 0, n_aln = 0, n_match = 0;
    long long win_ticks = 0, walk_ticks = 0; int n_windows = 0, n_general = 0;
    long long dbg_why = 0, dbg_a = 0, dbg_b = 0;      // (dead end of the walk: where and on what, for AlnOut.seg under ABPOA_HIP_DBG bit 8)
    if (spec_static && role == 0) {      // clear the tables, then let the helpers go (whatever the row loop's status: they wait for this)
        for (int t = lane; t < (SPEC_WK - 1) * SPEC_PM_ROWS; t += 64) pm_all[t] = 0;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) ctl[0] = gen;
    }
    if (status == 0 && b.ret_cigar && d.cigar_cap < gn + qlen + 2) status = ABPOA_HIP_EBACKTRACK;      // (a walk emits at most one word per row or column it leaves: no per-step capacity test)
    if (role >= 1 && status != 0) { helper_out(-1); return; }
    if (status == 0 && b.ret_cigar) {
        const int lds0 = b.lds.phase_off + (role >= 1 ? role * half_lds : 0);
        DirBt &B = *(DirBt *)(lds_raw + lds0);
        unsigned char *win = lds_raw + lds0 + (int)sizeof(DirBt);
        const int win_bytes = half_lds - (int)sizeof(DirBt);
        GLOBAL_AS uint64_t *cg = vgpr_ptr(b.cigar + d.cigar_off + (role >= 1 ? (long long)role * d.cigar_cap : 0));      // (helper r's words: part r of the set's cigar slots)
        bool merged = false; int idx1 = 0, met = 0;      // met: the helper whose path this walk ran into; idx1: index of the cigar word of the cell where
        int t_next = (role > 0 ? role : 0) + 1;        // the next helper down whose table this walk may look its cells up in
        int w_lo = 1, w_hi = 0, w_cref = 0; bool w_tri = false;    // window = rows [w_lo, w_hi], empty at start; column origin of the AB values; column slices (not whole rows)?
        // ---- stage the window for a walk that stands at (hi, jtop); returns A of row hi
        unsigned nx_end = 0; bool nx_ok = false;      // whole-row windows: where the window below this one ends in the arena (= where this one starts)
        auto load_window = [&](int hi, int jtop) __attribute__((always_inline)) -> int {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            WG_SYNC();
            // The walk usually leaves a whole-row window through its lowest row: the next window then ends exactly where this one started, and its copy --
            // the win_bytes in front of that point, rows are adjacent in the arena -- can go out BEFORE the rows' geometry is known (which only says
            // which of the copied rows are complete): one memory round trip per window instead of two.
            bool early = nx_ok && hi == w_lo - 1;
            const unsigned e_lo = nx_end > (unsigned)win_bytes ? nx_end - (unsigned)win_bytes : 0u;
            if (early) {
                const int n16 = (int)((nx_end - e_lo) >> 4);
                const int4 *src = (const int4 *)(arena + e_lo); int4 *dst = (int4 *)win;
                for (int i0 = 0; i0 < n16; i0 += 64) {
                    const int idx = i0 + lane;
                    if (idx < n16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx), (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
                }
            }
            const int cref = jtop - 4096;                       // (below every staged column: A values stay in 16 bits)
            // candidates: lane l holds rows hi - l - 64 q (descending rows: cumulative sizes are plain prefix sums); every load of every candidate in flight together
            int bs[NQ], es[NQ]; long long co[NQ]; unsigned pdv[NQ], pdh[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = hi - 64 * q - lane, rc = imax(r, 1);
                bs[q] = g_bsn[rc]; es[q] = g_esn[rc]; co[q] = g_coff[rc]; pdv[q] = row_pd[2 * rc]; pdh[q] = row_pd[2 * rc + 1];
            }
            int W[NQ], pc[NQ]; unsigned sa[NQ]; bool vq[NQ];
#pragma unroll
            // (arena byte offsets fit 32 bits)
            for (int q = 0; q < NQ; ++q) { vq[q] = hi - 64 * q - lane >= 1; W[q] = (es[q] - bs[q] + 1) * PN; pc[q] = bs[q] * PN; sa[q] = (unsigned)(co[q] * S); }
            const unsigned end_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(sa[0] + (unsigned)(W[0] * DB)));
            // whole rows: rows are adjacent in the arena, [start of row r, end of row hi) must fit
            int r_full = 0;
#pragma unroll
            for (int q = 0; q < NQ; ++q) r_full += __builtin_popcountll(__ballot(vq[q] && (early ? sa[q] >= e_lo : end_hi - sa[q] <= (unsigned)win_bytes)));
            if (early && r_full == 0) {
                // the speculative copy holds no complete row (row hi -- its words plus, behind them, the score records of the row below when that one keeps
                // them -- is larger than the window: a spill row after a band blow-up, or the 64-row windows of four wavefronts on one walk): let the copy
                // drain, forget it, and size the window for (hi, jtop) as if nothing had been copied
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                early = false; nx_ok = false;
#pragma unroll
                for (int q = 0; q < NQ; ++q) r_full += __builtin_popcountll(__ballot(vq[q] && end_hi - sa[q] <= (unsigned)win_bytes));
            }
            const bool narrow = early || r_full >= imin(16, hi);
            int sl[NQ], ns[NQ], off[NQ], R;
            unsigned s_lo = 0;
            if (narrow) {
                R = r_full;
                const int ql = (R - 1) >> 6, ll = (R - 1) & 63;
#pragma unroll
                for (int q = 0; q < NQ; ++q) if (q == ql) s_lo = (unsigned)__builtin_amdgcn_readlane((int)sa[q], ll);
                nx_end = s_lo;                       // (the start of the lowest complete row)
                if (early) s_lo = e_lo;              // (the copy started in front of it)
#pragma unroll
                for (int q = 0; q < NQ; ++q) { sl[q] = pc[q]; ns[q] = W[q]; off[q] = (int)(sa[q] - s_lo); }
            } else {
                int run = 0; R = 0;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int tl = imax(pc[q], (jtop - 64 * q - lane - DIR_TRI_SLACK) & ~(ALIGN - 1)), th = imin(pc[q] + W[q], (jtop + ALIGN) & ~(ALIGN - 1));
                    sl[q] = tl; ns[q] = vq[q] ? imax(0, th - tl) : 0;
                    const int incl = run + wave_scan_add_i32(ns[q] * DB);
                    off[q] = incl - ns[q] * DB; run = __builtin_amdgcn_readlane(incl, 63);
                    R += __builtin_popcountll(__ballot(vq[q] && incl <= win_bytes));
                }
                if (R < 1) R = 1;
            }
            const int lo = hi - R + 1;
            // ---- the words: LDS-DMA, issued now, waited for at the end (the records are built meanwhile)
            nx_ok = narrow;
            if (narrow && early) {}
            else if (narrow) {
                const int n16 = (int)((end_hi - s_lo) >> 4);
                const int4 *src = (const int4 *)(arena + s_lo); int4 *dst = (int4 *)win;
                for (int i0 = 0; i0 < n16; i0 += 64) {
                    const int idx = i0 + lane;
                    if (idx < n16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx), (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const unsigned srcq = sa[q] + (unsigned)((sl[q] - pc[q]) * DB);
                    const int nrow = imin(64, R - 64 * q);
                    for (int u = 0; u < nrow; ++u) {
                        const int n_ = __builtin_amdgcn_readlane(ns[q], u) * DB / 16, ob = __builtin_amdgcn_readlane(off[q], u);
                        const unsigned so = (unsigned)__builtin_amdgcn_readlane((int)srcq, u);
                        const int4 *sp = (const int4 *)(arena + so); unsigned char *dp = win + ob;
                        for (int i0 = 0; i0 < n_; i0 += 64)
                            if (i0 + lane < n_) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + i0 + lane),
                                    (__attribute__((address_space(3))) void *)(dp + i0 * 16), 16, 0, 0);
                    }
                }
            }
            // ---- row records: A of every window row first, then each row's predecessors look theirs up
            int A[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                A[q] = off[q] - ((sl[q] - cref) << DBL);
                const int li = hi - 64 * q - lane - lo;
                if (vq[q] && li >= 0) { B.rowA[li] = A[q]; B.pd[li] = make_int2((int)pdv[q], (int)pdh[q]); B.stg[li] = (sl[q] & 0x7fff) | ((sl[q] > pc[q] ? 1 : 0) << 15) | (ns[q] << 16); }
            }
            WG_SYNC();
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int r = hi - 64 * q - lane, li = r - lo;
                if (!(vq[q] && li >= 0)) continue;
                int pk[2];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const int dk = (int)((pdv[q] >> (8 * k)) & 0xffu), pr = r - dk; const bool in = dk != 255 && pr >= lo;
                    pk[k] = in ? (dk | (((B.rowA[pr - lo] >> 1) & 0xffff) << 8)) : 255;      // (AB is even: kept halved, windows of up to 64 KB)
                }
                B.rec[li] = make_int2(pk[0], pk[1]);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WG_SYNC();
            w_lo = lo; w_hi = hi; w_cref = cref;
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
            w_tri = !narrow;
            return __builtin_amdgcn_readfirstlane(A[0]);
        };
        // ---- cigar: a word is stored once, when the next one starts (an insertion run keeps growing in `last_word` until then); words carry the ROW
        //      where the reference has the node id -- the final pass below replaces it
        uint64_t last_word = 0; bool have_pending = false;
        auto store_word = [&](int idx, uint64_t wv) __attribute__((always_inline)) {      // (every lane stores the same value to the same address: one transaction, no exec juggling)
            GLOBAL_AS uint64_t *p_ = cg + idx;
            const int lo_ = (int)(wv & 0xffffffffull), hi_ = (int)(wv >> 32);
            asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p_), "v"(make_int2(lo_, hi_)) : "memory");
        };
        auto push = [&](int op, int len, int node_row, int query_id) __attribute__((always_inline)) {      // reference abpoa_align.h:54-73
            const uint64_t L = (uint64_t)(int64_t)len;
            if (n_cigar == 0 || op != ABPOA_HIP_CINS || op != (int)(last_word & 0xf)) {
                if (have_pending) store_word(n_cigar - 1, last_word);
                const uint64_t n_id = (uint64_t)(int64_t)node_row, q_id = (uint64_t)(int64_t)query_id;
                if (op == ABPOA_HIP_CMATCH) last_word = n_id << 34 | q_id << 4 | (uint64_t)op;
                else if (op == ABPOA_HIP_CINS) last_word = q_id << 34 | L << 4 | (uint64_t)op;
                else last_word = n_id << 34 | L << 4 | (uint64_t)op;
                have_pending = true; ++n_cigar;
            } else last_word += L << 4;
        };
        // a run of match steps: rows in `runv` (lane = step), first step at column run_j; its words are made and stored here, a lane each
        int runv = 0, run_n = 0, run_j = 0;
        auto flush_run = [&]() __attribute__((always_inline)) {
            if (run_n == 0) return;
            if (have_pending) { store_word(n_cigar - 1, last_word); have_pending = false; }
            if (spec) {      // the cells of this run: looked up in the table of the next helper down, noted in this helper's own
                const int row_ = lane < run_n ? runv : -1, col_ = run_j - lane;
                const int row_first = __builtin_amdgcn_readlane(runv, 0);      // (the highest row of the run)
                while (t_next < SPEC_WK && row_first < start_row(t_next) - SPEC_PM_ROWS + 1) ++t_next;      // (the whole run is below that table)
                if (t_next < SPEC_WK) {
                    const int tl_ = row_ - (start_row(t_next) - SPEC_PM_ROWS + 1);
                    const bool in2 = row_ >= 0 && (unsigned)tl_ < (unsigned)SPEC_PM_ROWS;
                    const int e_ = in2 ? ((lds_vint_t *)(pm_all + (t_next - 1) * SPEC_PM_ROWS))[tl_] : 0;
                    const unsigned long long hit_ = __ballot(in2 && e_ != 0 && (e_ & 0xffff) == col_);
                    if (hit_) { const int f_ = __builtin_ctzll(hit_); idx1 = (int)((unsigned)__builtin_amdgcn_readlane(e_, f_) >> 16); met = t_next; run_n = f_; merged = true; if (run_n == 0) return;
                            }
                }
                if (role >= 1) { const int t_ = row_ - R_lo; if (lane < run_n && (unsigned)t_ < (unsigned)SPEC_PM_ROWS) pm[t_] = col_ | ((n_cigar + lane) << 16); }
            }
            if (lane < run_n) cg[n_cigar + lane] = (uint64_t)(unsigned)runv << 34 | (uint64_t)(unsigned)(run_j - 1 - lane) << 4 | (uint64_t)ABPOA_HIP_CMATCH;
            // every cell a run passed must lie in its row's staged band (whole-row windows do not test it step by step; the rows of a run are all in the
            // current window: a run is flushed before the window changes): a cell outside is a dead end of the walk
            { const int st_ = B.stg[lane < run_n ? runv - w_lo : 0];
              if (__any(lane < run_n && (unsigned)(run_j - lane - (st_ & 0x7fff)) >= ((unsigned)st_ >> 16))) { status = ABPOA_HIP_EBACKTRACK; dbg_why = 1;
                      dbg_a = ((long long)run_n << 32) | (unsigned)run_j; dbg_b = ((long long)w_lo << 32) | (unsigned)w_hi; } }
            n_cigar += run_n; n_aln += run_n; bt_steps += run_n; run_n = 0;
            last_word = (uint64_t)ABPOA_HIP_CMATCH;            // (the last word so far is a match: the next insertion starts a word of its own)
        };

        int i = sgpr(best_i), j = sgpr(best_j), start_i = i, start_j = j, cur_op = OP_ALL, indel_first = role >= 1 ? 0 : 1, pend = 0;
        if (j < qlen && role < 1) push(ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
        const long long t_walk0 = (long long)__builtin_amdgcn_s_memtime();
        const int win_a = (int)(unsigned)(size_t)(lds_byte_t *)win, rec_a = (int)(unsigned)(size_t)(lds_byte_t *)(unsigned char *)B.rec;      // LDS byte addresses
        const int stg_a = (int)(unsigned)(size_t)(lds_byte_t *)(unsigned char *)B.stg;
        auto lds_w = [&](int addr) __attribute__((always_inline)) { return (int)*(const lds_dw_t *)(size_t)(unsigned)addr; };
        auto lds_i = [&](int addr) __attribute__((always_inline)) { return (int)*(const lds_i32_t *)(size_t)(unsigned)addr; };
        auto lds_r = [&](int addr) __attribute__((always_inline)) { const v2i_t v = *(const lds_i2_t *)(size_t)(unsigned)addr; return make_int2(v.x, v.y); };
        int Ai = 0; bool reloaded = false, restage = true;        // A of row i; restage: row i is not (known to be) in the window
        while (i > 0 && j > 0 && status == 0) {
            if (restage || i > w_hi || i < w_lo) { flush_run(); if (merged) break; Ai = load_window(i, j); restage = false; reloaded = true; }
            // ---- match run: while a match is what the reference tries first (M allowed, indel_first == 0) and the word names one of the first two predecessors
            if ((cur_op & OP_M) && indel_first == 0 && pend == 0) {
                if (run_n == 64) { flush_run(); if (merged) break; }
                if (run_n == 0) run_j = j;
                int recp = rec_a + ((i - w_lo) << 3), wb = win_a + ((j - w_cref) << DBL), budget = imin(64 - run_n, j);
                const int n0 = run_n;
                // (two copies of the loop: whole-row windows need no "is the cell staged" test -- every cell of a window row is, and a cell outside its row's
                //  band is caught when the run is flushed)
                auto run_loop = [&](auto tric) __attribute__((always_inline)) {
                    constexpr bool TRI = decltype(tric)::value;
                    do {
                        int2 rc_v = lds_r(recp); int w_v = lds_w(wb + Ai), st_v = 0;
                        if (TRI) st_v = lds_i(stg_a + ((recp - rec_a) >> 1));
                        asm volatile("" : "+v"(w_v), "+v"(rc_v.x), "+v"(rc_v.y), "+v"(st_v));
                        // not staged: the full step sorts it out
                        if (TRI) { const unsigned st = (unsigned)__builtin_amdgcn_readfirstlane(st_v); if ((unsigned)(j - (int)(st & 0x7fffu)) >= (st >> 16)) break; }
                        const unsigned k1 = ((unsigned)__builtin_amdgcn_readfirstlane(w_v) & 15u) - 1u;
                        if (k1 > 1u) break;                      // no match at this cell, or one through a later predecessor: the full step below
                        const unsigned long long pp = ((unsigned long long)(unsigned)rc_v.y << 32) | (unsigned long long)(unsigned)rc_v.x;
                        const unsigned pk = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(pp >> (k1 << 5)));
                        const int dk = (int)(pk & 0xffu);
                        if (dk == 255) break;                    // ... not in the window (or far away): the full step
                        { const int slot = sgpr(run_n), row_s = sgpr(i);
                          asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(runv) : "s"(row_s), "s"(slot) : "m0"); }
                        ++run_n; i -= dk; --j; recp -= dk << 3; wb -= DB; Ai = ((int)(pk << 8) >> 16) << 1;
                    } while (--budget != 0);
                };
                if (w_tri) run_loop(std::true_type{}); else run_loop(std::false_type{});
                const int i_prev = run_n != n0 ? __builtin_amdgcn_readlane(runv, (run_n - 1) & 63) : i;      // (the row of the run's last step)
                if (run_n != n0) { start_i = i_prev; start_j = j + 1; cur_op = OP_ALL; reloaded = false; }
                if (i <= 0 || j <= 0) continue;
                if (run_n == 64) continue;                       // (the run buffer is full: flush at the top, then on with the run)
            }
            // ---- full step: any state, the reference's priority order (:109-429) decided from the words (oracle/dir_model.c)
            flush_run(); ++n_general;
            if (status != 0 || merged) break;
            const int stw = __builtin_amdgcn_readfirstlane(B.stg[i - w_lo]);
            const int2 pd2 = uniform2(B.pd[i - w_lo]);
            const int sli = stw & 0x7fff, cut = (stw >> 15) & 1, nsi = (int)((unsigned)stw >> 16), si = j - sli;
            if ((unsigned)si >= (unsigned)nsi || (si == 0 && cut)) {      // the cell (or, possibly, its stored left neighbour) is not staged: re-centre the window on (i, j) once
                // ... it is not there: outside the row's band, no such cell
                if (reloaded) { status = ABPOA_HIP_EBACKTRACK; dbg_why = 2; dbg_a = ((long long)i << 32) | (unsigned)j; dbg_b = ((long long)stw << 32) | (unsigned)Ai; break; }
                restage = true; continue;
            }
            reloaded = false;
            const int waddr = win_a + Ai + ((j - w_cref) << DBL);
            int w_v = lds_w(waddr), wl_v = lds_w(si > 0 ? waddr - DB : waddr);
            asm volatile("" : "+v"(w_v), "+v"(wl_v));
            const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(w_v), wl = si > 0 ? (unsigned)__builtin_amdgcn_readfirstlane(wl_v) : 0u;      // (wl == 0: column j - 1 is not stored)
            const int kM = (int)(w & 15u);
            int kE[2], uE[2], dF[2], lF[2];
            if (GAP == 1) { kE[0] = (w >> DIRA_KE1_SH) & 15; uE[0] = (w >> DIRA_UE1_SH) & 7; dF[0] = (w >> DIRA_DF1_SH) & 7; lF[0] = (w >> DIRA_LF1_SH) & 3; kE[1] = 0; uE[1] = 0; dF[1] = 0;
                    lF[1] = 0; }
            else { kE[0] = (w >> DIRC_KE1_SH) & 15; kE[1] = (w >> DIRC_KE2_SH) & 15; uE[0] = (w >> DIRC_UE1_SH) & 7; uE[1] = (w >> DIRC_UE2_SH) & 31;
                   dF[0] = (w >> DIRC_DF1_SH) & 7; dF[1] = (w >> DIRC_DF2_SH) & 31; lF[0] = (w >> DIRC_LF1_SH) & 3; lF[1] = (w >> DIRC_LF2_SH) & 3; }
            if (pend) { cur_op = uE[pend - 1] == 0 ? (OP_M | OP_F) : (pend == 1 ? OP_E1 : OP_E2); pend = 0; }      // the deletion that led here: was this cell's E opened from its H? (reference :200)
            start_i = i; start_j = j; ++bt_steps;
            int hit = 0;
            // move to predecessor k (0-based list index) of row i: from the record, else (more than four predecessors / a far one) from the CSR arrays
            auto go_pred = [&](int k) __attribute__((always_inline)) -> bool {
                const int dk = k < 8 ? (int)(((unsigned)(k < 4 ? pd2.x : pd2.y) >> (8 * (k & 3))) & 0xffu) : 255;
                if (dk != 255) { i -= dk; if (i >= w_lo) Ai = __builtin_amdgcn_readfirstlane(B.rowA[i - w_lo]); else restage = true; return true; }
                const int po = __builtin_amdgcn_readfirstlane(gld_i32(pred_off + i)), po1 = __builtin_amdgcn_readfirstlane(gld_i32(pred_off + i + 1));
                if (k >= po1 - po) return false;
                i = __builtin_amdgcn_readfirstlane(gld_i32(pred_row + po + k));
                if (i >= w_lo && i <= w_hi) Ai = __builtin_amdgcn_readfirstlane(B.rowA[i - w_lo]); else restage = true;      // (usually still inside the window)
                return true;
            };
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (kM == 0) return;
                const int row_ = i;
                if (!go_pred(kM - 1)) return;
                cur_op = OP_ALL; hit = 1;
                push(ABPOA_HIP_CMATCH, 1, row_, j - 1);
                --j; ++n_aln;
                if (set_indel) indel_first = 0;
            };
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M; int kk0 = 64, kk1 = 64;
                if ((cur_op & OP_E1) && kE[0] >= 1 && (!viaM || uE[0] == o1)) kk0 = kE[0];
                if (GAP == 2 && (cur_op & OP_E2) && kE[1] >= 1 && (!viaM || uE[1] == o2)) kk1 = kE[1];
                if (kk0 != 64 || kk1 != 64) {                    // first predecessor in list order, E1 before E2 for the same one
                    const bool use1 = kk0 <= kk1; const int row_ = i;
                    if (go_pred((use1 ? kk0 : kk1) - 1)) {
                        cur_op = use1 ? OP_E1 : OP_E2; pend = use1 ? 1 : 2;      // (M|F instead if the predecessor's E was opened from its H: decided when its word is read)
                        hit = 1; push(ABPOA_HIP_CDEL, 1, row_, j - 1);
                    }
                }
            }
            if (!hit && (cur_op & OP_F)) {
                for (int x = 0; x < (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 0 ? OP_F1 : OP_F2, ox = x == 0 ? o1 : o2;
                    if (!(cur_op & bit)) continue;
                    if ((cur_op & OP_M) && dF[x] != 0) continue;                 // H == F
                    if (wl == 0u) continue;                                       // column j - 1 is not stored
                    int lit = lF[x];
                    if (lit == DIR_LIT_NONE) {
                        int kMl, uEl0, uEl1, dFl;
                        if (GAP == 1) { kMl = wl & 15; uEl0 = 0; uEl1 = 0; dFl = (wl >> DIRA_DF1_SH) & 7; }
                        else { kMl = wl & 15; uEl0 = (wl >> DIRC_UE1_SH) & 7; uEl1 = (wl >> DIRC_UE2_SH) & 31; dFl = x == 0 ? (wl >> DIRC_DF1_SH) & 7 : (wl >> DIRC_DF2_SH) & 31; }
                        const int h_is_hv = kMl != 0 || (GAP == 2 && (uEl0 == o1 || uEl1 == o2));
                        // the plane cannot decide this one (dir_plane.h); dbg bit 9: at the first insertion decided from the words (tests of the redo path)
                        if ((!h_is_hv && dFl > ox) || (b.dbg & 512)) { status = ABPOA_HIP_STATUS_NEED_SCORES; break; }
                        lit = dir_f_origin(h_is_hv, dFl, ox);
                    }
                    if (lit == DIR_LIT_OPEN) { cur_op = OP_M | OP_E; hit = 1; }
                    else if (lit == DIR_LIT_EXT) { cur_op = bit; hit = 1; }
                }
                if (status != 0) break;
                if (hit) { push(ABPOA_HIP_CINS, 1, i, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            if (!hit && status == 0) { status = ABPOA_HIP_EBACKTRACK; dbg_why = 3 | ((long long)cur_op << 8) | ((long long)indel_first << 16) | ((long long)n_general << 32);
                    dbg_a = ((long long)i << 32) | (unsigned)j; dbg_b = ((long long)w << 32) | (unsigned)pd2.x; }
        }
        walk_ticks = (long long)__builtin_amdgcn_s_memtime() - t_walk0;
        if (status == 0) {
            flush_run();
            if (!merged) { if (j > 0) push(ABPOA_HIP_CINS, j, -1, j - 1); if (have_pending) store_word(n_cigar - 1, last_word); }
        }
        if (role >= 1) {      // helper: publish where its walk ended (or whom it met) and leave (the main wavefront writes the result)
            WG_SYNC();
            if (lane == 0) { my[2] = n_cigar; my[3] = j; my[4] = start_i; my[5] = start_j; my[6] = bt_steps; my[7] = (status == 0 && merged) ? ((met << 16) | idx1) : -1; }
            helper_out(status);
            return;
        }
        // the chain of walks: segment q = words [seg_off[q], seg_off[q] + seg_len[q]) of wavefront seg_wk[q]
        int seg_wk[SPEC_WK], seg_off[SPEC_WK], seg_len[SPEC_WK], n_seg = 1;
        seg_wk[0] = 0; seg_off[0] = 0; seg_len[0] = n_cigar;
#pragma unroll
        for (int q = 1; q < SPEC_WK; ++q) { seg_wk[q] = 0; seg_off[q] = 0; seg_len[q] = 0; }
        if (status == 0 && merged) {      // the rest is the helpers': wait for each in turn, take over the end state of the last
            int nxt = met, from = idx1;
#pragma unroll
            for (int q = 1; q < SPEC_WK; ++q) {
                if (nxt <= 0 || status != 0) continue;
                lds_vint_t *h = ctl + 8 * nxt;
                while (h[0] != gen) __builtin_amdgcn_s_sleep(4);
                const int hs = h[1];
                if (hs != 0) { status = hs; continue; }      // (the helper met it on the common path: so would this walk have)
                seg_wk[q] = nxt; seg_off[q] = from; seg_len[q] = h[2] - from; n_seg = q + 1;
                j = h[3]; start_i = h[4]; start_j = h[5]; bt_steps += h[6];
                const int mm = h[7];
                if (mm >= 0) { nxt = mm >> 16; from = mm & 0xffff; } else nxt = 0;
            }
            if (status == 0) { n_cigar = 0;
#pragma unroll
                for (int q = 0; q < SPEC_WK; ++q) n_cigar += seg_len[q]; }
        }
        if (status == 0) {
            n_aln = best_j - j;      // (every match and insertion step takes one query base, a deletion none)
            WG_SYNC();
            // ---- final pass, a lane per word: row -> node id, matched bases, reversal (reference abpoa_reverse_cigar, abpoa_align.h:88-96)
            GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
            auto fix = [&](uint64_t wv, int &nm) __attribute__((always_inline)) -> uint64_t {
                const int op = (int)(wv & 0xf);
                if (op == ABPOA_HIP_CINS) return wv;
                const int row_ = (int)(wv >> 34);
                // (the query straight from HBM: this backtrack needs it nowhere else)
                if (op == ABPOA_HIP_CMATCH) { const int q = (int)((wv >> 4) & 0x3fffffffu); nm += (int)row_base[row_] == (int)g_query[q] ? 1 : 0; }
                return (wv & 0x3ffffffffull) | ((uint64_t)(int64_t)row_node_id[row_] << 34);
            };
            int nm = 0;
            const int half = n_cigar >> 1;
            static_assert(SPEC_WK == 4, "the chain below is written out for four segments");
            const int c0_ = seg_len[0], c1_ = c0_ + seg_len[1], c2_ = c1_ + seg_len[2];      // (segments that do not exist are empty)
            const long long p1_ = (long long)seg_wk[1] * d.cigar_cap + seg_off[1] - c0_, p2_ = (long long)seg_wk[2] * d.cigar_cap + seg_off[2] - c1_,
                    p3_ = (long long)seg_wk[3] * d.cigar_cap + seg_off[3] - c2_;
            auto rd = [&](int k_) __attribute__((always_inline)) -> uint64_t {      // (a chain of walks: the words of the helpers follow this wavefront's)
                const long long at_ = k_ < c0_ ? (long long)k_ : (k_ < c1_ ? p1_ + k_ : (k_ < c2_ ? p2_ + k_ : p3_ + k_));
                return cg[at_];
            };
            for (int k = lane; k < half; k += 128) {      // (two pairs of words per lane and turn: their loads -- the words, then node id and base by row -- overlap)
                const int k2 = k + 64; const bool v2 = k2 < half;
                uint64_t wa = rd(k), wc = rd(n_cigar - 1 - k), wa2 = v2 ? rd(k2) : 0, wc2 = v2 ? rd(n_cigar - 1 - k2) : 0;
                const uint64_t a = fix(wa, nm), c_ = fix(wc, nm);
                if (b.rev_cigar) { cg[k] = a; cg[n_cigar - 1 - k] = c_; } else { cg[k] = c_; cg[n_cigar - 1 - k] = a; }
                if (v2) { const uint64_t a2 = fix(wa2, nm), c2 = fix(wc2, nm);
                          if (b.rev_cigar) { cg[k2] = a2; cg[n_cigar - 1 - k2] = c2; } else { cg[k2] = c2; cg[n_cigar - 1 - k2] = a2; } }
            }
            if ((n_cigar & 1) && lane == 0) cg[half] = fix(rd(half), nm);
            n_match = __builtin_amdgcn_readlane(wave_scan_add_i32(nm), 63);
            node_e = row_node_id[best_i]; query_e = best_j - 1;
            node_s = row_node_id[start_i]; query_s = start_j - 1;
        }
    }
    if (lane == 0) {
        AlnOut o; for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = ts.seg[i_];
        o.status = status; o.best_score = best_score; o.best_row = best_i; o.best_col = best_j;
        o.node_s = node_s; o.node_e = node_e; o.query_s = query_s; o.query_e = query_e;
        o.n_aln_bases = n_aln; o.n_matched_bases = n_match; o.n_cigar = n_cigar; o.pad = -DB;      // pad < 0: direction-plane arena (bytes per word)
        o.n_cells = ts.n_cells; o.cells_used = ts.cursor;
        if (b.dbg & 256) { o.seg[0] = dbg_why; o.seg[1] = dbg_a; o.seg[2] = dbg_b; o.seg[3] = ((long long)best_i << 32) | (unsigned)best_j; o.seg[4] = n_cigar; o.seg[5] = bt_steps; }
        else if (!(b.dbg & 128)) { o.seg[5] = win_ticks; o.seg[4] = (long long)n_windows * 1000; o.seg[3] = (long long)n_general * 1000; o.seg[0] = 0; o.seg[1] = walk_ticks;
                o.seg[2] = (long long)bt_steps * 1000; }
        o.clk_dp = ts.clk1 - ts.clk0; o.clk_bt = (long long)__builtin_amdgcn_s_memtime() - ts.clk1; o.n_rows_done = ts.rows_done; o.n_bt_steps = bt_steps;
        *out_rec = o;
    }
}

}  // namespace abpoa_hip
