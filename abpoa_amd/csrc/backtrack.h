#pragma once
#include "dp_common.h"

namespace abpoa_hip {

// Everything after the row loop: global best (reference :1028-1041), backtrack (:109-429) and the result record.  Shared by the
// general kernel and the fast-loop kernel; reads only what the row loops left in HBM (planes, per-row band geometry).
struct TailState { long long cursor, n_cells, clk0, clk1, seg[6]; int status, rows_done, best_score, best_i, best_j; };

// CW = 0: plane-major arena rows (general kernel); CW > 0: cell records of CW values (fast loop, see rows_fast)
template <typename T, int GAP, int CW = 0>
__device__ __forceinline__ void finish_alignment(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec, const TailState &ts) {
    constexpr int PN = Width<T>::PN;
    constexpr int P = CW > 0 ? CW : (GAP == 0 ? 1 : (GAP == 1 ? 3 : 5));      // values per column in the arena
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    constexpr int PL_FLAG = GAP == 0 ? 1 : (GAP == 1 ? 3 : (sizeof(T) == 2 ? 6 : 5));      // cell records only: the row loop's match flag (0 = not known)
    const int lane = threadIdx.x & 63;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m;
    const bool local = b.align_mode == ABPOA_HIP_LOCAL_MODE;
    const bool banded = b.wb >= 0;
    const T e1 = (T)b.e1, oe1 = (T)(b.o1 + b.e1), e2 = (T)b.e2, oe2 = (T)(b.o2 + b.e2);
    GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    const bool q_in_lds = qlen <= b.lds.q_cap;
    auto dp_end_of = [&](int row, int end_sn_row) __attribute__((always_inline)) { return (banded || row == 0) ? (end_sn_row + 1) * PN - 1 : qlen; };
    int status = ts.status, best_score = ts.best_score, best_i = ts.best_i, best_j = ts.best_j, bt_steps = 0;
    const long long cursor = ts.cursor, n_cells = ts.n_cells, clk0 = ts.clk0, clk1 = ts.clk1; const int rows_done = ts.rows_done;
    const long long *seg = ts.seg;
    WG_SYNC();       // all of this wave's plane / band stores have landed before the loads below

    // ------------------------------------------------------------------ global best, reference :1028-1041
    if (status == 0 && b.align_mode == ABPOA_HIP_GLOBAL_MODE) {
        for (int k = pred_off[gn - 1]; k < pred_off[gn]; ++k) {
            int in_row = pred_row[k];
            int pe = g_esn[in_row], pb = g_bsn[in_row];
            int dpe = dp_end_of(in_row, pe);
            int end = qlen > dpe ? dpe : qlen;
            int score = (int)planes[g_coff[in_row] + (long long)(end - pb * PN) * (CW > 0 ? CW : 1)];
            if (score > best_score) { best_score = score; best_i = in_row; best_j = end; }
        }
    }

    // ------------------------------------------------------------------ backtrack, reference :109-429
    // The walk is executed redundantly (uniformly) by all lanes so that the LDS window of the arena can be
    // refilled cooperatively; only lane 0 writes cigar words.
    int n_cigar = 0, node_s = 0, node_e = 0, query_s = 0, query_e = 0, n_aln = 0, n_match = 0;
    long long bt_win_ticks = 0, bt_n_windows = 0, bt_slow_steps = 0, bt_wa = 0, bt_wb = 0, bt_flag_steps = 0;
    if (status == 0 && b.ret_cigar) {
        BtLds &B = *(BtLds *)(lds_raw + b.lds.phase_off);
        T *bt = (T *)(lds_raw + b.lds.phase_off + b.lds.bt_off);
        const long long bt_cells = (CW > 0 ? b.lds.bt_bytes_tail : b.lds.bt_bytes) / (int)sizeof(T);      // (the tail kernel's window is sized on its own: four of its workgroups per CU)
        int bt_lo = 1, bt_hi = 0, bt_pbase = 0, bt_margin = 0;   // window = rows [bt_lo, bt_hi], empty at start
        long long bt_c0 = 0;                                     // arena cell of B.coff[0]
        GLOBAL_AS uint64_t *cg = vgpr_ptr(b.cigar + d.cigar_off);
        const int cap = d.cigar_cap;
        const bool cap_safe = cap >= gn + qlen + 2;               // a walk emits at most one word per row or column it leaves: no per-step capacity check needed
        uint64_t last_word = 0;
        long long win_ticks = 0, win_a = 0; int n_windows = 0;
        auto load_window = [&](int hi) __attribute__((always_inline)) {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            WG_SYNC();
            int lo = imax(0, hi - BTR + 1);
            const int r = lo + lane;
            int my_b = -1, my_e = -1; long long my_c = 0;
            if (r <= hi) { my_b = g_bsn[r]; my_e = g_esn[r]; my_c = g_coff[r]; }
            // end of row hi = its offset + P * width (never-computed rows carry zero width)
            const int hb = g_bsn[hi], he = g_esn[hi];
            const long long c_end = g_coff[hi] + (hb >= 0 ? (long long)(he - hb + 1) * PN * P : 0);
            // smallest lo' whose segment [coff[lo'], c_end) fits the LDS tile
            const bool fits = (r <= hi) && (c_end - my_c) <= bt_cells;
            const unsigned long long mk = __ballot(fits);
            const int sh = mk ? __builtin_ctzll(mk) : (hi - lo);     // worst case: a single row (may still not fit -> HBM path)
            lo += sh;
            if (r >= lo && r <= hi) {
                const int i = r - lo;
                B.bsn[i] = my_b; B.esn[i] = my_e; B.coff[i] = my_c;
                B.poff[i] = pred_off[r]; B.nid[i] = row_node_id[r]; B.base[i] = row_base[r];
            }
            if (lane == 0) { B.coff[hi - lo + 1] = c_end; B.poff[hi - lo + 1] = pred_off[hi + 1]; }
            WG_SYNC();
            bt_lo = lo; bt_hi = hi; bt_c0 = B.coff[0]; bt_pbase = B.poff[0]; bt_margin = imin(4, (hi - lo) / 2);
            const int pn_t = imin(BTP, B.poff[hi - lo + 1] - bt_pbase);
            for (int i = lane; i < pn_t; i += 64) B.pred[i] = pred_row[bt_pbase + i];
            long long ncell = c_end - bt_c0; if (ncell > bt_cells) ncell = 0;        // does not fit: leave the tile empty
            if (ncell == 0) { bt_hi = bt_lo - 1; }
            // 16-byte coalesced copy (arena offsets are multiples of PN cells = 32 bytes)
            const int4 *src = (const int4 *)(planes + bt_c0); int4 *dst = (int4 *)bt;
            const int n16 = (int)(ncell * (long long)sizeof(T) / 16);
            for (int i0 = 0; i0 < n16; i0 += 64 * 8) {                            // 8 loads in flight per lane, then 8 LDS stores
                int4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int idx = i0 + u * 64 + lane; v[u] = idx < n16 ? src[idx] : make_int4(0, 0, 0, 0); }
#pragma unroll
                for (int u = 0; u < 8; ++u) { const int idx = i0 + u * 64 + lane; if (idx < n16) dst[idx] = v[u]; }
            }
            WG_SYNC();
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
        };
        // ---- window of the lane-parallel walk (cell-record arenas): rows [hi - R + 1, hi] x columns [jtop - WC + 1, jtop].  The walk
        //      moves up-left (each step: a predecessor row and / or one column back), so a column SLICE of every row is enough;
        //      with 10 kb reads a whole row is 4-8 KB and whole-row staging would hold 3-6 rows.  Leaving the slice (a long
        //      insertion run) simply reloads the window at the current cell.
        int win_i = -1, win_j = -1;                               // cell the current window was loaded for
        bool win_narrow = false;                                  // the window holds whole rows (every cell of a window row is staged)
        auto load_window_cols = [&](int hi, int jtop) __attribute__((always_inline)) {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            WG_SYNC();
            const int max_rec = (int)(bt_cells / (CW > 0 ? CW : 1));
            // columns of a slice: a path advances one column per step but 1.5-2.5 rows (bubbles of the graph), and the window holds max_rec records,
            // so slices are kept narrow enough for ~30 rows (measured on 10 kb reads: 32 columns with 16-byte records, 16 with 32-byte records;
            // diagnostic override ABPOA_HIP_BT_WC)
            int WC = b.lds.bt_wc;
            if (WC <= 0) {      // WC x R = max_rec with R / WC = rows per step: ~1.2 (5 % reads: 16-byte records here), ~2.4 (15 % reads: the convex default, 32-byte records)
                const float rho = (CW * (int)sizeof(T) >= 32) ? 2.4f : 1.2f;
                WC = imax(16, imin(64, ((int)__builtin_sqrtf((float)max_rec / rho) + 4) & ~7));
            }
            // candidate rows: the 64 rows ending at hi (lane = row - lo64); how many of them are staged is decided below
            const int lo64 = imax(0, hi - BTR + 1), n64 = hi - lo64 + 1;
            const int r = lo64 + lane; const bool rv64 = lane < n64;
            int b_ = -1, e_ = -1, po = 0, po1 = 0, nid_ = 0, bs_ = 0; long long c_ = 0;
            if (rv64) { b_ = g_bsn[r]; e_ = g_esn[r]; c_ = g_coff[r]; po = pred_off[r]; po1 = pred_off[r + 1]; nid_ = row_node_id[r]; bs_ = row_base[r]; }
            const int pbc = b_ >= 0 ? b_ * PN : 0, W = b_ >= 0 ? (e_ - b_ + 1) * PN : 0;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const long long tw1 = (long long)__builtin_amdgcn_s_memtime(); win_a += tw1 - tw0;
            // whole rows if at least 16 of them fit (narrow bands; rows are adjacent in the arena -> one contiguous copy), else slices
            const int Wrev = __builtin_amdgcn_ds_bpermute((n64 - 1 - lane) << 2, rv64 ? W : 0x100000);      // lane l <- row hi - l
            const int cum = wave_scan_add_i32(lane < n64 ? Wrev : 0x100000);
            const int r_full = __builtin_popcountll(__ballot(cum <= max_rec));
            const bool narrow = r_full >= imin(16, n64);
            // (8-byte records: a slice starts and ends on an even column -- it is up to two columns wider than WC, and the rows of a window must leave room for
            //  that: sized by WC alone, 48 rows of 34 records overran a 12 KB window by 96 records and the topmost row read back zeros -- EBACKTRACK on ragged reads
            //  in the wide loop; the retry passes hid it)
            // (linear gaps: records of 2 / 4 bytes -- slices start and end on multiples of 8 / 4 columns)
            constexpr int EVEN = (CW > 0 && CW * (int)sizeof(T) < 16) ? 16 / (CW * (int)sizeof(T)) - 1 : 0;
            const int R = narrow ? r_full : imin(n64, imax(4, max_rec / (WC + 2 * EVEN)));
            const int lo = hi - R + 1, nrow = R, li = lane - (lo - lo64);           // li: index of this lane's row inside the window
            const bool rv = rv64 && li >= 0;
            // (8-byte records: a slice starts and ends on an even column, so that it is a whole number of 16-byte pieces at a 16-byte address -- the band
            //  starts on a multiple of PN columns and is a multiple of PN wide)
            const int sl = narrow ? pbc : imax(pbc, (jtop - WC + 1) & ~EVEN), sh = narrow ? pbc + W : imin(pbc + W, (jtop + 1 + EVEN) & ~EVEN), ns = rv ? imax(0, sh - sl) : 0;
            const int incl = wave_scan_add_i32(ns);
            const int off_rec = incl - ns;
            const int pbase = __builtin_amdgcn_readlane(po, lo - lo64);
            const int pn_t = imin(BTP, __builtin_amdgcn_readlane(po1, n64 - 1) - pbase);
            if (rv) {
                // n_pred 255: not for the lane-parallel steps
                B.rinfo[li] = make_int4(pbc | (W << 16), off_rec * CW, ((po - pbase) & 0xffff) | (((po1 - po > 64 || po1 - pbase > BTP) ? 255 : po1 - po) << 16) | (bs_ << 24), nid_);
                B.rinfo2[li] = sl | (ns << 16);
                B.srcoff[li] = c_ + (long long)(sl - pbc) * CW;
            }
            WG_SYNC();
            const long long tw1b = (long long)__builtin_amdgcn_s_memtime(); win_a += tw1b - tw1;       // (debug split: scan + LDS tables)
            // the window's predecessor rows travel with the cell copy below (issued here, waited for with the first batch of cells)
            int prv[BTP / 64];
#pragma unroll
            for (int k_ = 0; k_ < BTP / 64; ++k_) { const int e_ = k_ * 64 + lane; gld_async(prv[k_], (const int32_t *)pred_row + pbase + (e_ < pn_t ? e_ : 0)); }
            // staged records: 8 rows per batch, lane = column inside the slice
            // narrow bands: every slice is a whole row, and the rows are adjacent in the arena -> one contiguous 16-byte-wide copy
            if (narrow) {
                const int l0 = lo - lo64;
                const long long c_lo = (long long)(unsigned)__builtin_amdgcn_readlane((int)(c_ & 0xffffffffll), l0) | ((long long)__builtin_amdgcn_readlane((int)(c_ >> 32), l0) << 32);
                const int n16 = (int)((long long)__builtin_amdgcn_readlane(incl, 63) * CW * (int)sizeof(T) / 16);
                const int4 *src = (const int4 *)(planes + c_lo); int4 *dst = (int4 *)bt;
                // LDS-DMA (global_load_lds_dwordx4: 64 lanes x 16 bytes land at a wave-uniform LDS base + 16 * lane): the whole window is in flight in
                // one HBM round trip and passes through no vector registers
                // (M0 carries the LDS base; tools/probes/glds_high_lds.hip: it reaches all of the workgroup's LDS, also above 64 KB)
                for (int i0 = 0; i0 < n16; i0 += 64) {
                    const int idx = i0 + lane;
                    if (idx < n16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx), (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
                }
            } else {
                // slices: one LDS-DMA per row and 1 KB of slice (lane = 16-byte piece of the slice; a slice is contiguous in the arena and in the window);
                // every row of the window in flight together, no vector registers; the per-row constants travel by v_readlane, not through LDS
                constexpr int REC = (int)(CW * sizeof(T));               // bytes per cell record: 8, 16 or 32 (8: slices start and end on even columns, below)
                const int offb = off_rec * REC; const long long srcv = c_ + (long long)(sl - pbc) * CW;
                const int src_lo = (int)(srcv & 0xffffffffll), src_hi = (int)(srcv >> 32);
                for (int u = 0; u < nrow; ++u) {
                    const int rr = u + (lo - lo64);
                    const int np16 = __builtin_amdgcn_readlane(ns, rr) * REC / 16, ob = __builtin_amdgcn_readlane(offb, rr);
                    const long long so = (long long)(unsigned)__builtin_amdgcn_readlane(src_lo, rr) | ((long long)__builtin_amdgcn_readlane(src_hi, rr) << 32);
                    const int4 *src = (const int4 *)(planes + so); unsigned char *dstb = (unsigned char *)bt + ob;
                    for (int i0 = 0; i0 < np16; i0 += 64)
                        if (i0 + lane < np16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i0 + lane),
                                (__attribute__((address_space(3))) void *)(dstb + i0 * 16), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            gld_wait();                                     // (a window without cells to copy still has predecessor rows in flight)
#pragma unroll
            for (int k_ = 0; k_ < BTP / 64; ++k_) {
                const int e = k_ * 64 + lane; if (e >= pn_t) continue;
                const int pr_ = prv[k_]; const bool ok = pr_ >= lo && pr_ <= hi;
                const int4 ri_ = B.rinfo[ok ? pr_ - lo : 0];
                B.edge[e] = make_int4(pr_, ok ? ri_.x : 0, ri_.y, ok ? 1 : 0); B.edge2[e] = make_int4(ri_.z, ri_.w, B.rinfo2[ok ? pr_ - lo : 0], 0);      // (not staged: empty band)
            }
            WG_SYNC();
            bt_lo = lo; bt_hi = hi; bt_pbase = pbase; win_i = hi; win_j = jtop; win_narrow = narrow;
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
        };
        // cigar words are collected 64 at a time in a VGPR pair (lane = word index & 63) and written out as one coalesced store per 64
        // words: a store per step would sit in the memory pipeline when the next step's LDS reads are issued, and the compiler's
        // s_waitcnt vmcnt(0) in front of those reads then costs a full HBM write round trip per step
        int cgw_lo = 0, cgw_hi = 0;
        auto flush_cigar = [&](int base, int n) __attribute__((always_inline)) {
            if (lane < n) cg[base + lane] = ((uint64_t)(unsigned)cgw_hi << 32) | (uint64_t)(unsigned)cgw_lo;
        };
        auto push = [&](int op, int len, int node_id, int query_id) __attribute__((always_inline)) {      // reference abpoa_align.h:54-73
            uint64_t L = (uint64_t)(int64_t)len;
            if (n_cigar == 0 || op != ABPOA_HIP_CINS || op != (int)(last_word & 0xf)) {
                if (n_cigar >= cap) { status = ABPOA_HIP_EBACKTRACK; return; }
                if (n_cigar > 0 && (n_cigar & 63) == 0) flush_cigar(n_cigar - 64, 64);      // the previous 64 words are final now
                uint64_t n_id = (uint64_t)(int64_t)node_id, q_id = (uint64_t)(int64_t)query_id, wv;
                if (op == ABPOA_HIP_CMATCH) wv = n_id << 34 | q_id << 4 | (uint64_t)op;
                else if (op == ABPOA_HIP_CINS) wv = q_id << 34 | L << 4 | (uint64_t)op;
                else wv = n_id << 34 | L << 4 | (uint64_t)op;
                last_word = wv; ++n_cigar;
            } else last_word += L << 4;
            const int w_lo = sgpr((int)(last_word & 0xffffffffull)), w_hi = sgpr((int)(last_word >> 32)), w_idx = sgpr((n_cigar - 1) & 63);
            asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0"
                         : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx) : "m0");
        };
        struct Geo { int pb, pe; long long off; bool in_tile; };
        auto geo_of = [&](int row_) __attribute__((always_inline)) {
            Geo g;
            g.in_tile = CW == 0 && row_ >= bt_lo && row_ <= bt_hi;      // (cell-record arenas: the LDS window holds column slices for the lane-parallel walk only)
            const int i = g.in_tile ? row_ - bt_lo : 0;
            g.pb = B.bsn[i]; g.pe = B.esn[i]; g.off = B.coff[i] - bt_c0;
            if (!g.in_tile) { g.pb = gld_i32(g_bsn + row_); g.pe = gld_i32(g_esn + row_); g.off = gld_i64(g_coff + row_); }
            return g;
        };
        auto cell = [&](const Geo &g, int plane, int col_) __attribute__((always_inline)) -> int {
            const long long Wp = (long long)(g.pe - g.pb + 1) * PN;
            const long long idx = CW > 0 ? g.off + (long long)(col_ - g.pb * PN) * CW + plane : g.off + plane * Wp + (col_ - g.pb * PN);
            int v = (int)bt[g.in_tile ? idx : 0];
            if (!g.in_tile) v = gld_cell((GLOBAL_AS const T *)(planes + idx));
            return v;
        };
        auto in_range = [&](const Geo &g, int row_, int col_) __attribute__((always_inline)) { return col_ >= g.pb * PN && col_ <= dp_end_of(row_, g.pe); };
        auto stored = [&](const Geo &g, int col_) __attribute__((always_inline)) { return col_ >= g.pb * PN && col_ <= (g.pe + 1) * PN - 1; };
        auto qcode = [&](int j_) __attribute__((always_inline)) { int v = (int)s_query[q_in_lds ? j_ : 0]; if (!q_in_lds) v = gld_u8(g_query + j_); return v; };

        int i = best_i, j = best_j, start_i = best_i, start_j = best_j, cur_op = OP_ALL, indel_first = 1;
        if (best_j < qlen) push(ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
        // ---- lane-parallel step (cell-record arenas, i.e. the fast path): one LDS round trip each for (a) the row's own record,
        //      (b) its predecessor list, (c) the predecessors' geometry, (d) every score the decision can need -- lane k holds
        //      predecessor k -- then the reference's priority order (:109-429) is evaluated on ballot masks.  Falls through to
        //      the one-read-at-a-time walk below whenever a predecessor is outside the staged window or the row has > 64 of them.
        bool bt_walk_narrow = true;                                // false once a window had to be staged as column slices
        // the walk's state is the same in every lane; values that came out of vector loads (best cell, the one-read-at-a-time step) are
        // moved to SGPRs so that the step loops below run on scalar branches
        auto uniformize = [&]() __attribute__((always_inline)) {
            i = sgpr(i); j = sgpr(j); cur_op = sgpr(cur_op); indel_first = sgpr(indel_first); status = sgpr(status); n_cigar = sgpr(n_cigar);
            n_aln = sgpr(n_aln); n_match = sgpr(n_match); start_i = sgpr(start_i); start_j = sgpr(start_j); bt_steps = sgpr(bt_steps);
            bt_lo = sgpr(bt_lo); bt_hi = sgpr(bt_hi); bt_pbase = sgpr(bt_pbase); win_i = sgpr(win_i); win_j = sgpr(win_j);
            last_word = (uint64_t)(unsigned)sgpr((int)(last_word & 0xffffffffull)) | ((uint64_t)(unsigned)sgpr((int)(last_word >> 32)) << 32);
        };
        uniformize();
        const long long t_walk0 = (long long)__builtin_amdgcn_s_memtime();
        bool local_done = false;                                   // local mode: the walk reached a cell with H == 0 (reference :126)
        do {      // fast steps; one slow step whenever a fast one cannot be taken; back to fast steps
        // ---- lane-parallel step (cell-record arenas, i.e. the fast path).  The current row's record is carried in SGPRs; round
        //      trip 1 fetches its predecessor edge records (lane k = predecessor k), its own cells and the query code, round trip 2
        //      the predecessors' cells and the substitution score; the reference's priority order (:109-429) is then evaluated on
        //      ballot masks and the chosen predecessor's record becomes the current one.  Falls through to ONE step of the
        //      one-read-at-a-time walk below whenever a predecessor is outside the staged window or the row has > 64 of them.
        int4 cr = make_int4(0, 0, 0, 0); int cr2 = 0, cr_row = -1;               // rinfo / rinfo2 of row cr_row
        // two copies of the step loop: whole-row windows (narrow bands: no slice bookkeeping at all) and column-slice windows
        while (CW > 0 && i > 0 && j > 0 && status == 0 && bt_walk_narrow && !local_done) {
            if (i > bt_hi || i < bt_lo) { load_window_cols(i, j); cr_row = -1; if (!win_narrow) { bt_walk_narrow = false; break; } }
            if (cr_row != i) { cr = uniform4(B.rinfo[i - bt_lo]); cr_row = i; }
            // ---- match run.  The row loop left "1 + index of the first predecessor whose diagonal cell gives H" in every cell record it
            //      wrote on its straight-line path (0 = not known).  While a match is what the reference tries first (:130-160 with M allowed
            //      and indel_first == 0) and the flag is set, a step is ONE LDS round trip (flag, query code, the row's edge records) and a
            //      handful of scalar instructions; anything else leaves the loop for the full step below.
            if ((cur_op & OP_M) && indel_first == 0 && q_in_lds && cap_safe) {
                int mi_ = i, mj = j, pi_ = i, nm_v = 0, w_lo = 0, w_hi = 0; int4 mc = cr;
                const int nc0 = n_cigar;
                int slots = n_cigar == 0 ? 64 : ((64 - (n_cigar & 63)) & 63);           // words that still fit before the VGPR pair has to be written out
                for (;;) {
                    const int si_ = mj - (mc.x & 0xffff), np_ = (mc.z >> 16) & 0xff;
                    if ((unsigned)si_ >= ((unsigned)mc.x >> 16)) break;                  // (cannot happen on a sane path: the cell lies in its row's band)
                    int fl_v = (int)bt[mc.y + si_ * CW + PL_FLAG];
                    int qc_v = (int)s_query[mj - 1];
                    const int e_idx = (mc.z & 0xffff) + (lane < np_ ? lane : 0);        // (n_pred 255 = row not eligible: the reads stay inside the LDS image, the result is not used)
                    int4 er = B.edge[e_idx & (BTP - 1)]; int4 er2 = B.edge2[e_idx & (BTP - 1)];
                    // this step's cigar word, lane slot and flush test need nothing from the loads: computed while they are in flight
                    const int w_lo_n = sgpr(((mj - 1) << 4) | ABPOA_HIP_CMATCH), w_hi_n = sgpr(mc.w << 2), w_idx_n = sgpr(n_cigar & 63), bs_n = sgpr((int)((unsigned)mc.z >> 24));
                    asm volatile("" :: "s"(w_lo_n), "s"(w_hi_n), "s"(w_idx_n), "s"(bs_n));
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(fl_v), "+v"(qc_v), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er2.x), "+v"(er2.y));      // every load issued before the one wait
                    const int fl = __builtin_amdgcn_readfirstlane(fl_v);
                    const int ks = (fl - 1) & 63, ery = __builtin_amdgcn_readlane(er.y, ks);
                    // flag known and in range, row eligible, column j-1 inside that predecessor's band (empty when it is not staged)
                    if (!((unsigned)(fl - 1) < (unsigned)np_ && np_ != 255 && (unsigned)(mj - 1 - (ery & 0xffff)) < ((unsigned)ery >> 16))) break;
                    if (slots == 0) { flush_cigar(n_cigar - 64, 64); slots = 64; }
                    --slots;
                    w_lo = w_lo_n; w_hi = w_hi_n;                                                                 // node id << 34 | query index << 4 | op
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx_n) : "m0");
                    ++n_cigar; nm_v += (qc_v == bs_n) ? 1 : 0;
                    pi_ = mi_; --mj;
                    mi_ = __builtin_amdgcn_readlane(er.x, ks);
                    mc = make_int4(ery, __builtin_amdgcn_readlane(er.z, ks), __builtin_amdgcn_readlane(er2.x, ks), __builtin_amdgcn_readlane(er2.y, ks));
                    if (imin(mi_, mj) <= 0) break;
                }
                const int moved = n_cigar - nc0;
                if (moved) {
                    start_i = pi_; start_j = mj + 1; bt_steps += moved; bt_flag_steps += moved; n_aln += moved; n_match += __builtin_amdgcn_readfirstlane(nm_v);
                    last_word = ((uint64_t)(unsigned)w_hi << 32) | (uint64_t)(unsigned)w_lo; cur_op = OP_ALL;
                    i = mi_; j = mj; cr = mc; cr_row = i;
                    if (i <= 0 || j <= 0) continue;
                }
            }
            const int pbi = cr.x & 0xffff, Wi = (int)((unsigned)cr.x >> 16), offi = cr.y;
            const int eb = cr.z & 0xffff, np = (cr.z >> 16) & 0xff, bs_ = (int)((unsigned)cr.z >> 24), id = cr.w;
            const int sli = pbi, nsi = Wi;                                      // whole rows are staged
            if (np > 64 || eb + np > BTP) break;
            // round trip 1: predecessor edge records, own cells, query code
            const int4 er = B.edge[eb + (lane < np ? lane : 0)]; const int4 er2 = B.edge2[eb + (lane < np ? lane : 0)];
            const int xi = j - pbi, si = j - sli;
            const bool st_jm1 = xi - 1 >= 0 && xi - 1 < Wi;                      // stored(gi, j-1)
            bool need = false;
            const T *ri = bt + offi + ((unsigned)si < (unsigned)nsi ? si : 0) * CW;
            // (every lane reads the same cell: keep the walk's state in SGPRs)
            const int Hij = __builtin_amdgcn_readfirstlane((int)ri[0]), E1ij = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E1]) : 0, E2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E2]) : 0,
                    F1ij = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F1]) : 0, F2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F2]) : 0;
            const T *rim1 = bt + offi + ((st_jm1 && si - 1 >= 0) ? si - 1 : 0) * CW;
            const int Hijm1 = __builtin_amdgcn_readfirstlane((int)rim1[0]), F1ijm1 = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F1]) : 0,
                    F2ijm1 = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F2]) : 0;
            if (local && Hij == 0) { local_done = true; break; }            // reference :126: the local walk ends on a zero cell
            const int qc = __builtin_amdgcn_readfirstlane(qcode(j - 1));
            const bool act = lane < np;
            // round trip 2: the predecessors' cells (lane k = predecessor k) and the substitution score
            const int pbk = er.y & 0xffff, Wk = (int)((unsigned)er.y >> 16), xk = j - pbk;
            const int sk = xk, nsk = Wk;
            const bool in_j = act && (unsigned)xk < (unsigned)Wk, in_jm1 = act && (unsigned)(xk - 1) < (unsigned)Wk;
            need = __any(act && er.w == 0);
            if (need) {                                                          // something this step reads is not staged: re-centre the window on (i, j) once
                if (win_i == i && win_j == j) break;
                load_window_cols(i, j); cr_row = -1; if (!win_narrow) { bt_walk_narrow = false; break; } continue;
            }
            const T *rk = bt + er.z + (in_j ? sk : 0) * CW, *rkm1 = bt + er.z + (in_jm1 ? sk - 1 : 0) * CW;
            const int Hk_j = (int)rk[0], E1k_j = GAP != 0 ? (int)rk[PL_E1] : 0, E2k_j = GAP == 2 ? (int)rk[PL_E2] : 0, Hk_jm1 = (int)rkm1[0];
            const int sc_ = s_mat[m * bs_ + qc];
            start_i = i; start_j = j; ++bt_steps;
            const unsigned long long mA = __ballot(in_jm1 && Hk_jm1 + sc_ == Hij);
            int hit = 0, k_sel = -1;
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (!mA) return;
                k_sel = __builtin_ctzll(mA);
                cur_op = OP_ALL; hit = 1;
                push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                --j; ++n_aln; n_match += (bs_ == qc);
                if (set_indel) indel_first = 0;
            };
            if constexpr (GAP == 0) {      // linear gaps, reference :109-190: match (unless an indel is tried first), deletion from the first predecessor that fits, insertion, match
                if (indel_first == 0) do_match(0);
                if (!hit) { const unsigned long long mD = __ballot(in_j && Hk_j - (int)e1 == Hij); if (mD) { k_sel = __builtin_ctzll(mD); hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1); } }
                if (!hit && st_jm1 && Hijm1 - (int)e1 == Hij) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; hit = 1; }
                if (!hit && indel_first == 1) do_match(1);
            } else {
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M;
                unsigned long long m1 = 0, m2 = 0;
                if (cur_op & OP_E1) m1 = __ballot(in_j && (viaM ? Hij == E1k_j : E1ij == E1k_j - (int)e1));
                if (GAP == 2 && (cur_op & OP_E2)) m2 = __ballot(in_j && (viaM ? Hij == E2k_j : E2ij == E2k_j - (int)e2));
                if (m1 | m2) {                                                   // first predecessor in list order, E1 before E2 for the same one
                    const int k1 = m1 ? __builtin_ctzll(m1) : 64, k2 = m2 ? __builtin_ctzll(m2) : 64;
                    const bool use1 = k1 <= k2; k_sel = use1 ? k1 : k2;
                    const unsigned long long mD = __ballot(in_j && (use1 ? Hk_j - (int)oe1 == E1k_j : Hk_j - (int)oe2 == E2k_j));
                    cur_op = ((mD >> k_sel) & 1) ? (OP_M | OP_F) : (use1 ? OP_E1 : OP_E2);
                    hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1);
                }
            }
            if (!hit && (cur_op & OP_F)) {
                for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 1 ? OP_F1 : OP_F2;
                    const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                    if (!(cur_op & bit)) continue;
                    const int Fij = x == 1 ? F1ij : F2ij;
                    if (!(cur_op & OP_M) || Hij == Fij) {
                        if (st_jm1) {
                            if (Hijm1 - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                            else if ((x == 1 ? F1ijm1 : F2ijm1) - ex == Fij) { cur_op = bit; hit = 1; }
                        }
                    }
                }
                if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            }
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
            if (k_sel >= 0) {                                                    // move to the chosen predecessor: its record comes along
                i = __builtin_amdgcn_readlane(er.x, k_sel);
                cr = make_int4(__builtin_amdgcn_readlane(er.y, k_sel), __builtin_amdgcn_readlane(er.z, k_sel), __builtin_amdgcn_readlane(er2.x, k_sel), __builtin_amdgcn_readlane(er2.y, k_sel));
                cr_row = i;
            }
        }
        // (the whole-row loop above carries no cr2: a walk that leaves it for a slow step -- a predecessor that is not staged even in the window centred on the
        //  cell -- passes through here first, and with a stale cr2 the cell index below fell on column 0 of the row, which in local mode reads as the zero
        //  cell that ends the walk: found with reads of ragged ends, whose local alignments reach predecessors hundreds of rows away)
        cr_row = -1;
        while (CW > 0 && i > 0 && j > 0 && status == 0 && !local_done) {
            if (i > bt_hi || i < bt_lo) { load_window_cols(i, j); cr_row = -1; }
            if (cr_row != i) { cr = uniform4(B.rinfo[i - bt_lo]); cr2 = __builtin_amdgcn_readfirstlane(B.rinfo2[i - bt_lo]); cr_row = i; }
            // ---- match run, as in the whole-row loop above; here a row's staged cells are the column slice cr2 = first column | count << 16
            if ((cur_op & OP_M) && indel_first == 0 && q_in_lds && cap_safe) {
                int mi_ = i, mj = j, pi_ = i, nm_v = 0, w_lo = 0, w_hi = 0; int4 mc = cr; int mc2 = cr2;
                const int nc0 = n_cigar;
                int slots = n_cigar == 0 ? 64 : ((64 - (n_cigar & 63)) & 63);           // words that still fit before the VGPR pair has to be written out
                for (;;) {
                    const int si_ = mj - (mc2 & 0xffff), np_ = (mc.z >> 16) & 0xff;
                    if ((unsigned)si_ >= ((unsigned)mc2 >> 16)) break;                   // the cell is outside the staged slice of its row: the full step re-centres the window
                    int fl_v = (int)bt[mc.y + si_ * CW + PL_FLAG];
                    int qc_v = (int)s_query[mj - 1];
                    const int e_idx = (mc.z & 0xffff) + (lane < np_ ? lane : 0);        // (n_pred 255 = row not eligible: the reads stay inside the LDS image, the result is not used)
                    int4 er = B.edge[e_idx & (BTP - 1)]; int4 er2 = B.edge2[e_idx & (BTP - 1)];
                    // (as in the whole-row loop)
                    const int w_lo_n = sgpr(((mj - 1) << 4) | ABPOA_HIP_CMATCH), w_hi_n = sgpr(mc.w << 2), w_idx_n = sgpr(n_cigar & 63), bs_n = sgpr((int)((unsigned)mc.z >> 24));
                    asm volatile("" :: "s"(w_lo_n), "s"(w_hi_n), "s"(w_idx_n), "s"(bs_n));
                    __builtin_amdgcn_sched_barrier(0);
                    asm volatile("" : "+v"(fl_v), "+v"(qc_v), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er2.x), "+v"(er2.y), "+v"(er2.z));      // every load issued before the one wait
                    const int fl = __builtin_amdgcn_readfirstlane(fl_v);
                    const int ks = (fl - 1) & 63, ery = __builtin_amdgcn_readlane(er.y, ks);
                    // flag known and in range, row eligible, column j-1 inside that predecessor's band (empty when it is not staged)
                    if (!((unsigned)(fl - 1) < (unsigned)np_ && np_ != 255 && (unsigned)(mj - 1 - (ery & 0xffff)) < ((unsigned)ery >> 16))) break;
                    if (slots == 0) { flush_cigar(n_cigar - 64, 64); slots = 64; }
                    --slots;
                    w_lo = w_lo_n; w_hi = w_hi_n;                                                                 // node id << 34 | query index << 4 | op
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 3\n\tv_writelane_b32 %0, %2, m0\n\tv_writelane_b32 %1, %3, m0" : "+v"(cgw_lo), "+v"(cgw_hi) : "s"(w_lo), "s"(w_hi), "s"(w_idx_n) : "m0");
                    ++n_cigar; nm_v += (qc_v == bs_n) ? 1 : 0;
                    pi_ = mi_; --mj;
                    mi_ = __builtin_amdgcn_readlane(er.x, ks);
                    mc = make_int4(ery, __builtin_amdgcn_readlane(er.z, ks), __builtin_amdgcn_readlane(er2.x, ks), __builtin_amdgcn_readlane(er2.y, ks)); mc2 = __builtin_amdgcn_readlane(er2.z, ks);
                    if (imin(mi_, mj) <= 0) break;
                }
                const int moved = n_cigar - nc0;
                if (moved) {
                    start_i = pi_; start_j = mj + 1; bt_steps += moved; bt_flag_steps += moved; n_aln += moved; n_match += __builtin_amdgcn_readfirstlane(nm_v);
                    last_word = ((uint64_t)(unsigned)w_hi << 32) | (uint64_t)(unsigned)w_lo; cur_op = OP_ALL;
                    i = mi_; j = mj; cr = mc; cr2 = mc2; cr_row = i;
                    if (i <= 0 || j <= 0) continue;
                }
            }
            const int pbi = cr.x & 0xffff, Wi = (int)((unsigned)cr.x >> 16), offi = cr.y;
            const int eb = cr.z & 0xffff, np = (cr.z >> 16) & 0xff, bs_ = (int)((unsigned)cr.z >> 24), id = cr.w;
            const int sli = cr2 & 0xffff, nsi = (int)((unsigned)cr2 >> 16);
            if (np > 64 || eb + np > BTP) break;
            // round trip 1: predecessor edge records, own cells, query code
            const int4 er = B.edge[eb + (lane < np ? lane : 0)]; const int4 er2 = B.edge2[eb + (lane < np ? lane : 0)];
            const int xi = j - pbi, si = j - sli;
            const bool st_jm1 = xi - 1 >= 0 && xi - 1 < Wi;                      // stored(gi, j-1)
            bool need = !win_narrow && ((unsigned)si >= (unsigned)nsi || (st_jm1 && si - 1 < 0));     // a cell of the own row outside the staged slice
            const T *ri = bt + offi + ((unsigned)si < (unsigned)nsi ? si : 0) * CW;
            // (every lane reads the same cell: keep the walk's state in SGPRs)
            const int Hij = __builtin_amdgcn_readfirstlane((int)ri[0]), E1ij = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E1]) : 0, E2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_E2]) : 0,
                    F1ij = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F1]) : 0, F2ij = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)ri[PL_F2]) : 0;
            const T *rim1 = bt + offi + ((st_jm1 && si - 1 >= 0) ? si - 1 : 0) * CW;
            const int Hijm1 = __builtin_amdgcn_readfirstlane((int)rim1[0]), F1ijm1 = GAP != 0 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F1]) : 0,
                    F2ijm1 = GAP == 2 ? __builtin_amdgcn_readfirstlane((int)rim1[PL_F2]) : 0;
            // reference :126: the local walk ends on a zero cell (a cell outside the staged slice is re-read after the reload below)
            if (local && Hij == 0 && (win_narrow || (unsigned)si < (unsigned)nsi)) { local_done = true; break; }
            const int qc = __builtin_amdgcn_readfirstlane(qcode(j - 1));
            const bool act = lane < np;
            // round trip 2: the predecessors' cells (lane k = predecessor k) and the substitution score
            const int pbk = er.y & 0xffff, Wk = (int)((unsigned)er.y >> 16), xk = j - pbk;
            const int slk = er2.z & 0xffff, nsk = (int)((unsigned)er2.z >> 16), sk = j - slk;
            const bool in_j = act && (unsigned)xk < (unsigned)Wk, in_jm1 = act && (unsigned)(xk - 1) < (unsigned)Wk;
            const bool stg_j = (unsigned)sk < (unsigned)nsk, stg_jm1 = (unsigned)(sk - 1) < (unsigned)nsk;
            need = need || (win_narrow ? __any(act && er.w == 0) : __any(act && (er.w == 0 || (in_j && !stg_j) || (in_jm1 && !stg_jm1))));
            if (need) {                                                          // something this step reads is not staged: re-centre the window on (i, j) once
                if (win_i == i && win_j == j) break;
                load_window_cols(i, j); cr_row = -1; continue;
            }
            const T *rk = bt + er.z + (in_j ? sk : 0) * CW, *rkm1 = bt + er.z + (in_jm1 ? sk - 1 : 0) * CW;
            const int Hk_j = (int)rk[0], E1k_j = GAP != 0 ? (int)rk[PL_E1] : 0, E2k_j = GAP == 2 ? (int)rk[PL_E2] : 0, Hk_jm1 = (int)rkm1[0];
            const int sc_ = s_mat[m * bs_ + qc];
            start_i = i; start_j = j; ++bt_steps;
            const unsigned long long mA = __ballot(in_jm1 && Hk_jm1 + sc_ == Hij);
            int hit = 0, k_sel = -1;
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (!mA) return;
                k_sel = __builtin_ctzll(mA);
                cur_op = OP_ALL; hit = 1;
                push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                --j; ++n_aln; n_match += (bs_ == qc);
                if (set_indel) indel_first = 0;
            };
            if constexpr (GAP == 0) {      // linear gaps, reference :109-190: match (unless an indel is tried first), deletion from the first predecessor that fits, insertion, match
                if (indel_first == 0) do_match(0);
                if (!hit) { const unsigned long long mD = __ballot(in_j && Hk_j - (int)e1 == Hij); if (mD) { k_sel = __builtin_ctzll(mD); hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1); } }
                if (!hit && st_jm1 && Hijm1 - (int)e1 == Hij) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; hit = 1; }
                if (!hit && indel_first == 1) do_match(1);
            } else {
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M;
                unsigned long long m1 = 0, m2 = 0;
                if (cur_op & OP_E1) m1 = __ballot(in_j && (viaM ? Hij == E1k_j : E1ij == E1k_j - (int)e1));
                if (GAP == 2 && (cur_op & OP_E2)) m2 = __ballot(in_j && (viaM ? Hij == E2k_j : E2ij == E2k_j - (int)e2));
                if (m1 | m2) {                                                   // first predecessor in list order, E1 before E2 for the same one
                    const int k1 = m1 ? __builtin_ctzll(m1) : 64, k2 = m2 ? __builtin_ctzll(m2) : 64;
                    const bool use1 = k1 <= k2; k_sel = use1 ? k1 : k2;
                    const unsigned long long mD = __ballot(in_j && (use1 ? Hk_j - (int)oe1 == E1k_j : Hk_j - (int)oe2 == E2k_j));
                    cur_op = ((mD >> k_sel) & 1) ? (OP_M | OP_F) : (use1 ? OP_E1 : OP_E2);
                    hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1);
                }
            }
            if (!hit && (cur_op & OP_F)) {
                for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 1 ? OP_F1 : OP_F2;
                    const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                    if (!(cur_op & bit)) continue;
                    const int Fij = x == 1 ? F1ij : F2ij;
                    if (!(cur_op & OP_M) || Hij == Fij) {
                        if (st_jm1) {
                            if (Hijm1 - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                            else if ((x == 1 ? F1ijm1 : F2ijm1) - ex == Fij) { cur_op = bit; hit = 1; }
                        }
                    }
                }
                if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            }
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
            if (k_sel >= 0) {                                                    // move to the chosen predecessor: its record comes along
                i = __builtin_amdgcn_readlane(er.x, k_sel);
                cr = make_int4(__builtin_amdgcn_readlane(er.y, k_sel), __builtin_amdgcn_readlane(er.z, k_sel), __builtin_amdgcn_readlane(er2.x, k_sel), __builtin_amdgcn_readlane(er2.y, k_sel));
                cr2 = __builtin_amdgcn_readlane(er2.z, k_sel); cr_row = i;
            }
        }
        int slow_budget = CW > 0 ? 1 : INT_MAX;
        while (i > 0 && j > 0 && status == 0 && !local_done && slow_budget-- > 0) {
            ++bt_slow_steps;
            if (CW == 0 && ((i < bt_lo + bt_margin && bt_lo > 0) || i > bt_hi || i < bt_lo)) load_window(i);
            const Geo gi = geo_of(i);
            const int Hij = cell(gi, 0, j);
            if (local && Hij == 0) { local_done = true; break; }
            start_i = i; start_j = j; ++bt_steps;
            int ps, np, id, bs_;
            { const int t = gi.in_tile ? i - bt_lo : 0; ps = B.poff[t]; np = B.poff[t + 1] - ps; id = B.nid[t]; bs_ = B.base[t]; }
            if (!gi.in_tile) { ps = gld_i32(pred_off + i); np = gld_i32(pred_off + i + 1) - ps; id = gld_i32(row_node_id + i); bs_ = gld_u8(row_base + i); }
            auto pred_bt = [&](int idx) __attribute__((always_inline)) { const int t = idx - bt_pbase; const bool ok = gi.in_tile && t >= 0 && t < BTP; int v = B.pred[ok ? t : 0];
                    if (!ok) v = gld_i32(pred_row + idx); return v; };
            const int qc = qcode(j - 1);
            const int s = s_mat[m * bs_ + qc];
            const int is_match = bs_ == qc;
            int hit = 0;
            auto try_match = [&](int set_indel) __attribute__((always_inline)) {
                for (int k = 0; k < np; ++k) {
                    const int pr = pred_bt(ps + k);
                    const Geo gp = geo_of(pr);
                    if (!in_range(gp, pr, j - 1)) continue;
                    if (cell(gp, 0, j - 1) + s == Hij) {
                        cur_op = OP_ALL; hit = 1;
                        push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                        i = pr; --j; ++n_aln; n_match += is_match;
                        if (set_indel) indel_first = 0;
                        break;
                    }
                }
            };
            if (GAP == 0) {
                if (indel_first == 0) try_match(0);
                if (!hit) {
                    for (int k = 0; k < np; ++k) {
                        const int pr = pred_bt(ps + k);
                        const Geo gp = geo_of(pr);
                        if (!in_range(gp, pr, j)) continue;
                        if (cell(gp, 0, j) - (int)e1 == Hij) { push(ABPOA_HIP_CDEL, 1, id, j - 1); i = pr; hit = 1; break; }
                    }
                }
                if (!hit && stored(gi, j - 1) && cell(gi, 0, j - 1) - (int)e1 == Hij) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; hit = 1; ++n_aln; }
                if (!hit && indel_first == 1) try_match(1);
            } else {
                if ((cur_op & OP_M) && indel_first == 0) try_match(0);
                if (!hit && (cur_op & OP_E)) {
                    for (int k = 0; k < np && !hit; ++k) {
                        const int pr = pred_bt(ps + k);
                        const Geo gp = geo_of(pr);
                        if (!in_range(gp, pr, j)) continue;
                        for (int x = 1; x <= (GAP == 2 ? 2 : 1); ++x) {
                            const int bit = x == 1 ? OP_E1 : OP_E2, pl = x == 1 ? PL_E1 : PL_E2;
                            const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                            if (!(cur_op & bit)) continue;
                            const int preE = cell(gp, pl, j);
                            const bool ok = (cur_op & OP_M) ? (Hij == preE) : (cell(gi, pl, j) == preE - ex);
                            if (ok) {
                                cur_op = (cell(gp, 0, j) - oex == preE) ? (OP_M | OP_F) : bit;
                                hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1); i = pr; break;
                            }
                        }
                    }
                }
                if (!hit && (cur_op & OP_F)) {
                    for (int x = 1; x <= (GAP == 2 ? 2 : 1) && !hit; ++x) {
                        const int bit = x == 1 ? OP_F1 : OP_F2, pl = x == 1 ? PL_F1 : PL_F2;
                        const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                        if (!(cur_op & bit)) continue;
                        const int Fij = cell(gi, pl, j);
                        if (!(cur_op & OP_M) || Hij == Fij) {
                            if (stored(gi, j - 1)) {
                                if (cell(gi, 0, j - 1) - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                                else if (cell(gi, pl, j - 1) - ex == Fij) { cur_op = bit; hit = 1; }
                            }
                        }
                    }
                    if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
                }
                if (!hit && (cur_op & OP_M) && indel_first == 1) try_match(1);
            }
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
        }
        if (CW > 0) uniformize();
        } while (CW > 0 && i > 0 && j > 0 && status == 0 && !local_done);
        bt_win_ticks = win_ticks; bt_n_windows = n_windows; bt_wa = win_a; bt_wb = (long long)__builtin_amdgcn_s_memtime() - t_walk0;
        if (status == 0) {
            if (j > 0) push(ABPOA_HIP_CINS, j, -1, j - 1);
            if (n_cigar > 0) { const int base_ = ((n_cigar - 1) >> 6) << 6; flush_cigar(base_, n_cigar - base_); }
            WG_SYNC();
            if (!b.rev_cigar) for (int k = lane; k < n_cigar >> 1; k += 64) { uint64_t t = cg[k]; cg[k] = cg[n_cigar - 1 - k]; cg[n_cigar - 1 - k] = t; }
            node_e = row_node_id[best_i]; query_e = best_j - 1;
            node_s = row_node_id[start_i]; query_s = start_j - 1;
        }
    }
    if (lane == 0) {
        AlnOut o; for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = 0;
        o.status = status; o.best_score = best_score; o.best_row = best_i; o.best_col = best_j;
        o.node_s = node_s; o.node_e = node_e; o.query_s = query_s; o.query_e = query_e;
        o.n_aln_bases = n_aln; o.n_matched_bases = n_match; o.n_cigar = n_cigar; o.pad = CW;      // pad = arena cell stride (0: plane-major)
        o.n_cells = n_cells; o.cells_used = cursor;
        for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = seg[i_];
        // (dbg bit 7: keep the row loop's counters) backtrack: ticks spent staging arena windows, number of windows
        if (!(b.dbg & 128)) { o.seg[5] = bt_win_ticks; o.seg[4] = bt_n_windows * 1000; o.seg[3] = bt_slow_steps * 1000; o.seg[0] = bt_wa; o.seg[1] = bt_wb; o.seg[2] = bt_flag_steps * 1000; }
        o.clk_dp = clk1 - clk0; o.clk_bt = (long long)__builtin_amdgcn_s_memtime() - clk1; o.n_rows_done = rows_done; o.n_bt_steps = bt_steps;
        *out_rec = o;
    }
}

}  // namespace abpoa_hip
