#pragma once
#include "dp_common.h"
#include "dir_plane.h"
#include "rows_fast.h"      // DirFmt, FastFmt
#include "backtrack.h"      // TailState

namespace abpoa_hip {

// Global best + backtrack over DIRECTION-PLANE arenas (dir_plane.h): the row loops left one word per cell that records every comparison the
// reference backtrack (src/simd_abpoa_align.c:109-429) would make there, so the walk reads 2 / 4 bytes per step, never a score, and every kind of
// step -- match, deletion, insertion -- is the same single LDS round trip: the cell's word, the word of its left neighbour and the row's
// predecessor edge records (lane k = predecessor k, each carrying the predecessor's own row record), then scalar decisions in the reference's
// priority order.  oracle/dir_model.c is the CPU statement of exactly this walk (checked against the value-comparing backtrack on every golden).
//
// Windows: rows [lo, hi] of the arena staged in LDS by LDS-DMA, up to DBTR rows with their row / edge tables built once per window.
//   * narrow bands: whole rows, one contiguous copy (rows are adjacent in the arena; a row that also keeps its score records drags them along);
//   * wide bands: a TRIANGLE of column slices -- a match step goes one column back and at least one row up, a deletion only up, so from (hi, jtop)
//     the walk reaches columns [jtop - (hi - r), jtop] of row r unless insertions (one column back in the same row) push it further left: row r stages
//     DIR_TRI_SLACK more columns than that (a path advances 1.2 - 2.4 rows per column, so the slack grows on its own with the distance; the one
//     column an insertion step looks left is part of it).  A walk that does leave a slice re-centres the window on its cell.
// The one situation the plane cannot decide (dir_plane.h: F origin under an F-term H with dF > o) ends the walk with ABPOA_HIP_STATUS_NEED_SCORES and
// the host redoes that alignment with score records; oracle/dir_model.c counts it: 0 in 250 000 steps of noisy 1-5 kb reads.
constexpr int DBTR = 128;     // rows per window
constexpr int DBTP = 192;     // predecessor edges per window
constexpr int DIR_TRI_SLACK = 9;
struct __attribute__((aligned(16))) DirBt {
    // rinfo = {first band column | band columns << 16, LDS byte offset of the row's first STAGED word, edge index | n_pred << 16 | base << 24, node id}
    // rinfo2 = first staged column | staged columns << 16
    // edge = {predecessor row, its rinfo.x, its rinfo.y, inside the window?}; edge2 = {its rinfo.z, its rinfo.w, its rinfo2, -}
    int4 rinfo[DBTR]; int4 edge[DBTP]; int4 edge2[DBTP]; int32_t rinfo2[DBTR];
};

template <typename T, int GAP>
__device__ __forceinline__ void finish_alignment_dir(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec, const TailState &ts) {
    constexpr int PN = Width<T>::PN, CW = FastFmt<T, GAP>::CW, DB = DirFmt<T, GAP>::DB, S = (int)sizeof(T);
    constexpr int ALIGN = 16 / DB;                    // columns per 16-byte piece of a row of words
    typedef typename std::conditional<GAP == 1, uint16_t, uint32_t>::type DW;
    const int lane = threadIdx.x & 63;
    const int gn = d.n_rows, qlen = d.qlen;
    const int o1 = b.o1, o2 = b.o2;
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);
    const unsigned char *arena = (const unsigned char *)planes;
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int status = ts.status, best_score = ts.best_score, best_i = ts.best_i, best_j = ts.best_j, bt_steps = 0;
    WG_SYNC();       // all of this wave's arena / band stores have landed before the loads below

    // ------------------------------------------------------------------ global best, reference :1028-1041 (the sink's predecessors keep their score records)
    if (status == 0) {
        for (int k = pred_off[gn - 1]; k < pred_off[gn]; ++k) {
            const int in_row = pred_row[k];
            const int pe = g_esn[in_row], pb = g_bsn[in_row];
            const int dpe = (pe + 1) * PN - 1, end = qlen > dpe ? dpe : qlen;
            const int score = (int)planes[g_coff[in_row] + (long long)DirFmt<T, GAP>::units(pe - pb + 1) * PN + (long long)(end - pb * PN) * CW];
            if (score > best_score) { best_score = score; best_i = in_row; best_j = end; }
        }
    }

    int n_cigar = 0, node_s = 0, node_e = 0, query_s = 0, query_e = 0, n_aln = 0, n_match = 0;
    long long win_ticks = 0, walk_ticks = 0; int n_windows = 0;
    if (status == 0 && b.ret_cigar) {
        DirBt &B = *(DirBt *)(lds_raw + b.lds.phase_off);
        unsigned char *bt = lds_raw + b.lds.phase_off + b.lds.bt_off;
        const int bt_bytes = b.lds.bt_bytes_tail;
        GLOBAL_AS uint64_t *cg = vgpr_ptr(b.cigar + d.cigar_off);
        const int cap = d.cigar_cap;
        uint64_t last_word = 0;
        int bt_lo = 1, bt_hi = 0;                                 // window = rows [bt_lo, bt_hi], empty at start
        // ---- window of rows [hi - R + 1, hi] for a walk that stands at (hi, jtop)
        auto load_window = [&](int hi, int jtop) __attribute__((always_inline)) {
            const long long tw0 = (long long)__builtin_amdgcn_s_memtime(); ++n_windows;
            WG_SYNC();
            // candidates: lane l holds row hi - l ("a") and row hi - 64 - l ("b"): descending rows, so that cumulative sizes are plain prefix sums
            const int ra = hi - lane, rb = hi - 64 - lane; const bool va = ra >= 0, vb = rb >= 0;
            int ba = 0, ea = -1, poa = 0, po1a = 0, nida = 0, bsa = 0, bb = 0, eb_ = -1, pob = 0, po1b = 0, nidb = 0, bsb = 0; long long ca = 0, cb = 0;
            if (va) { ba = g_bsn[ra]; ea = g_esn[ra]; ca = g_coff[ra]; poa = pred_off[ra]; po1a = pred_off[ra + 1]; nida = row_node_id[ra]; bsa = row_base[ra]; }
            if (vb) { bb = g_bsn[rb]; eb_ = g_esn[rb]; cb = g_coff[rb]; pob = pred_off[rb]; po1b = pred_off[rb + 1]; nidb = row_node_id[rb]; bsb = row_base[rb]; }
            const int Wa = va ? (ea - ba + 1) * PN : 0, Wb = vb ? (eb_ - bb + 1) * PN : 0, pca = ba * PN, pcb = bb * PN;
            // arena byte offsets fit 32 bits (an arena is far below 4 GB)
            const unsigned sa = (unsigned)(ca * S), sb = (unsigned)(cb * S);
            const unsigned end_hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(sa + (unsigned)(Wa * DB)));
            const int pend_hi = __builtin_amdgcn_readfirstlane(po1a);
            // whole rows: rows are adjacent in the arena, [start of row r, end of row hi) must fit the window; edges of rows r .. hi must fit the table
            const bool fa = va && end_hi - sa <= (unsigned)bt_bytes && pend_hi - poa <= DBTP, fb = vb && end_hi - sb <= (unsigned)bt_bytes && pend_hi - pob <= DBTP;
            const int r_full = __builtin_popcountll(__ballot(fa)) + __builtin_popcountll(__ballot(fb));
            const bool narrow = r_full >= imin(16, hi + 1);
            // triangle of column slices (see the header comment); slices start and end on 16-byte pieces of the row
            const int tla = imax(pca, (jtop - lane - DIR_TRI_SLACK) & ~(ALIGN - 1)), tha = imin(pca + Wa, (jtop + ALIGN) & ~(ALIGN - 1));
            const int tlb = imax(pcb, (jtop - 64 - lane - DIR_TRI_SLACK) & ~(ALIGN - 1)), thb = imin(pcb + Wb, (jtop + ALIGN) & ~(ALIGN - 1));
            const int sla = narrow ? pca : tla, nsa = va ? (narrow ? Wa : imax(0, tha - tla)) : 0;
            const int slb = narrow ? pcb : tlb, nsb = vb ? (narrow ? Wb : imax(0, thb - tlb)) : 0;
            int R, offa, offb;                                    // rows in the window; LDS byte offset of each candidate row's staged words
            if (narrow) { R = r_full; }
            else {
                const int ia = wave_scan_add_i32(nsa * DB), ta = __builtin_amdgcn_readlane(ia, 63), ib = ta + wave_scan_add_i32(nsb * DB);
                const bool ga = va && ia <= bt_bytes && pend_hi - poa <= DBTP, gb = vb && ib <= bt_bytes && pend_hi - pob <= DBTP;
                R = __builtin_popcountll(__ballot(ga)) + __builtin_popcountll(__ballot(gb));
                if (R < 1) R = 1;                                  // (a single row's slice always fits: a few 16-byte pieces)
                offa = ia - nsa * DB; offb = ib - nsb * DB;
            }
            const int lo = hi - R + 1;
            if (narrow) {      // offsets relative to the start of row lo
                const unsigned s_lo = (unsigned)(R <= 64 ? __builtin_amdgcn_readlane((int)sa, (R - 1) & 63) : __builtin_amdgcn_readlane((int)sb, (R - 65) & 63));
                offa = (int)(sa - s_lo); offb = (int)(sb - s_lo);
            }
            const int pbase = R <= 64 ? __builtin_amdgcn_readlane(poa, (R - 1) & 63) : __builtin_amdgcn_readlane(pob, (R - 65) & 63);
            if (va && lane < R) {
                const int li = ra - lo;
                B.rinfo[li] = make_int4(pca | (Wa << 16), offa, ((poa - pbase) & 0xffff) | (imin(po1a - poa, 255) << 16) | (bsa << 24), nida);
                B.rinfo2[li] = sla | (nsa << 16);
            }
            if (vb && lane + 64 < R) {
                const int li = rb - lo;
                B.rinfo[li] = make_int4(pcb | (Wb << 16), offb, ((pob - pbase) & 0xffff) | (imin(po1b - pob, 255) << 16) | (bsb << 24), nidb);
                B.rinfo2[li] = slb | (nsb << 16);
            }
            // the window's predecessor rows (and the band of the ones outside it) travel with the word copy below
            const int pn_t = imin(DBTP, pend_hi - pbase);
            int prv[DBTP / 64];
#pragma unroll
            for (int k_ = 0; k_ < DBTP / 64; ++k_) { const int e_ = k_ * 64 + lane; gld_async(prv[k_], (const int32_t *)pred_row + pbase + (e_ < pn_t ? e_ : 0)); }
            if (narrow) {
                const unsigned s_lo = (unsigned)(R <= 64 ? __builtin_amdgcn_readlane((int)sa, (R - 1) & 63) : __builtin_amdgcn_readlane((int)sb, (R - 65) & 63));
                const int n16 = (int)((end_hi - s_lo) >> 4);
                const int4 *src = (const int4 *)(arena + s_lo); int4 *dst = (int4 *)bt;
                for (int i0 = 0; i0 < n16; i0 += 64) {
                    const int idx = i0 + lane;
                    if (idx < n16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + idx), (__attribute__((address_space(3))) void *)(dst + i0), 16, 0, 0);
                }
            } else {
                // one LDS-DMA per row and KB of slice; the per-row constants travel by v_readlane
                const unsigned srca = sa + (unsigned)((sla - pca) * DB), srcb = sb + (unsigned)((slb - pcb) * DB);
                for (int u = 0; u < R; ++u) {
                    const int ln = u & 63;
                    const int np16 = (u < 64 ? __builtin_amdgcn_readlane(nsa, ln) : __builtin_amdgcn_readlane(nsb, ln)) * DB / 16;
                    const int ob = u < 64 ? __builtin_amdgcn_readlane(offa, ln) : __builtin_amdgcn_readlane(offb, ln);
                    const unsigned so = (unsigned)(u < 64 ? __builtin_amdgcn_readlane((int)srca, ln) : __builtin_amdgcn_readlane((int)srcb, ln));
                    const int4 *src = (const int4 *)(arena + so); unsigned char *dstb = bt + ob;
                    for (int i0 = 0; i0 < np16; i0 += 64)
                        if (i0 + lane < np16) __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i0 + lane), (__attribute__((address_space(3))) void *)(dstb + i0 * 16), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            WG_SYNC();                                            // tables and words are in LDS
            // edge records: a predecessor inside the window brings its row record along; one outside brings its band (for the range test of a step
            // that leaves the window through it) from the per-row arrays
#pragma unroll
            for (int k_ = 0; k_ < DBTP / 64; ++k_) {
                const int e = k_ * 64 + lane; if (e >= pn_t) continue;
                const int pr_ = prv[k_]; const bool ok = pr_ >= lo && pr_ <= hi;
                int4 ri_ = B.rinfo[ok ? pr_ - lo : 0]; int ri2_ = B.rinfo2[ok ? pr_ - lo : 0];
                if (!ok) { const int pb_ = g_bsn[pr_], pe_ = g_esn[pr_]; ri_ = make_int4((pb_ * PN) | (((pe_ - pb_ + 1) * PN) << 16), 0, 0, 0); ri2_ = 0; }
                B.edge[e] = make_int4(pr_, ri_.x, ri_.y, ok ? 1 : 0); B.edge2[e] = make_int4(ri_.z, ri_.w, ri2_, 0);
            }
            WG_SYNC();
            bt_lo = lo; bt_hi = hi;
            win_ticks += (long long)__builtin_amdgcn_s_memtime() - tw0;
        };
        // cigar words are collected 64 at a time in a VGPR pair (lane = word index & 63) and written out as one coalesced store per 64 words (backtrack.h)
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

        int i = sgpr(best_i), j = sgpr(best_j), start_i = i, start_j = j, cur_op = OP_ALL, indel_first = 1, pend = 0;
        if (j < qlen) push(ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
        const long long t_walk0 = (long long)__builtin_amdgcn_s_memtime();
        int4 cr = make_int4(0, 0, 0, 0); int cr2 = 0, cr_row = -1; bool reloaded = false;
        while (i > 0 && j > 0 && status == 0) {
            if (i > bt_hi || i < bt_lo) { load_window(i, j); cr_row = -1; reloaded = true; }
            if (cr_row != i) { cr = uniform4(B.rinfo[i - bt_lo]); cr2 = __builtin_amdgcn_readfirstlane(B.rinfo2[i - bt_lo]); cr_row = i; }
            const int pbi = cr.x & 0xffff, Wi = (int)((unsigned)cr.x >> 16);
            const int eb = cr.z & 0xffff, np = (cr.z >> 16) & 0xff, bs_ = (int)((unsigned)cr.z >> 24), id = cr.w;
            const int sli = cr2 & 0xffff, nsi = (int)((unsigned)cr2 >> 16), si = j - sli;
            if ((unsigned)(j - pbi) >= (unsigned)Wi || np > DIR_K_MAX) { status = ABPOA_HIP_EBACKTRACK; break; }      // outside the row's band: no such cell
            if ((unsigned)si >= (unsigned)nsi || (si == 0 && j - 1 >= pbi)) {      // the cell (or its stored left neighbour) is not staged: re-centre the window on (i, j) once
                if (reloaded) { status = ABPOA_HIP_EBACKTRACK; break; }
                load_window(i, j); cr_row = -1; reloaded = true; continue;
            }
            reloaded = false;
            // ---- the step's one LDS round trip: the cell's word, its left neighbour's, the query code, the row's edge records (lane k = predecessor k)
            const DW *wp = (const DW *)(bt + cr.y) + si;
            int w_v = (int)wp[0], wl_v = (int)wp[si > 0 ? -1 : 0], qc_v = (int)s_query[j - 1];
            const int e_idx = eb + (lane < np ? lane : 0);
            int4 er = B.edge[e_idx < DBTP ? e_idx : 0], er2 = B.edge2[e_idx < DBTP ? e_idx : 0];
            asm volatile("" : "+v"(w_v), "+v"(wl_v), "+v"(qc_v), "+v"(er.x), "+v"(er.y), "+v"(er.z), "+v"(er.w), "+v"(er2.x), "+v"(er2.y), "+v"(er2.z));      // every load issued before the one wait
            const unsigned w = (unsigned)__builtin_amdgcn_readfirstlane(w_v);
            const int kM = (int)(w & 15u);
            int kE[2], uE[2], dF[2], lF[2];
            if (GAP == 1) { kE[0] = (w >> DIRA_KE1_SH) & 15; uE[0] = (w >> DIRA_UE1_SH) & 7; dF[0] = (w >> DIRA_DF1_SH) & 7; lF[0] = (w >> DIRA_LF1_SH) & 3; kE[1] = 0; uE[1] = 0; dF[1] = 0; lF[1] = 0; }
            else { kE[0] = (w >> DIRC_KE1_SH) & 15; kE[1] = (w >> DIRC_KE2_SH) & 15; uE[0] = (w >> DIRC_UE1_SH) & 7; uE[1] = (w >> DIRC_UE2_SH) & 31;
                   dF[0] = (w >> DIRC_DF1_SH) & 7; dF[1] = (w >> DIRC_DF2_SH) & 31; lF[0] = (w >> DIRC_LF1_SH) & 3; lF[1] = (w >> DIRC_LF2_SH) & 3; }
            if (pend) { cur_op = uE[pend - 1] == 0 ? (OP_M | OP_F) : (pend == 1 ? OP_E1 : OP_E2); pend = 0; }      // the deletion that led here: was this cell's E opened from its H? (reference :200)
            start_i = i; start_j = j; ++bt_steps;
            int hit = 0, k_sel = -1;
            auto in_pred_band = [&](int k, int col) __attribute__((always_inline)) { const int ery = __builtin_amdgcn_readlane(er.y, k); return (unsigned)(col - (ery & 0xffff)) < ((unsigned)ery >> 16); };
            auto do_match = [&](int set_indel) __attribute__((always_inline)) {
                if (kM >= 1 && kM <= np && eb + kM <= DBTP && in_pred_band(kM - 1, j - 1)) {
                    k_sel = kM - 1; cur_op = OP_ALL; hit = 1;
                    push(ABPOA_HIP_CMATCH, 1, id, j - 1);
                    n_match += (bs_ == __builtin_amdgcn_readfirstlane(qc_v)) ? 1 : 0; --j; ++n_aln;
                    if (set_indel) indel_first = 0;
                }
            };
            if ((cur_op & OP_M) && indel_first == 0) do_match(0);
            if (!hit && (cur_op & OP_E)) {
                const bool viaM = cur_op & OP_M; int kk0 = 64, kk1 = 64;
                if ((cur_op & OP_E1) && kE[0] >= 1 && kE[0] <= np && (!viaM || uE[0] == o1) && in_pred_band(kE[0] - 1, j)) kk0 = kE[0];
                if (GAP == 2 && (cur_op & OP_E2) && kE[1] >= 1 && kE[1] <= np && (!viaM || uE[1] == o2) && in_pred_band(kE[1] - 1, j)) kk1 = kE[1];
                if (kk0 != 64 || kk1 != 64) {                    // first predecessor in list order, E1 before E2 for the same one
                    const bool use1 = kk0 <= kk1; k_sel = (use1 ? kk0 : kk1) - 1;
                    cur_op = use1 ? OP_E1 : OP_E2; pend = use1 ? 1 : 2;      // (M|F instead if the predecessor's E was opened from its H: decided when its word is read)
                    hit = 1; push(ABPOA_HIP_CDEL, 1, id, j - 1);
                }
            }
            if (!hit && (cur_op & OP_F)) {
                const unsigned wl = (unsigned)__builtin_amdgcn_readfirstlane(wl_v);
                for (int x = 0; x < (GAP == 2 ? 2 : 1) && !hit; ++x) {
                    const int bit = x == 0 ? OP_F1 : OP_F2, ox = x == 0 ? o1 : o2;
                    if (!(cur_op & bit)) continue;
                    if ((cur_op & OP_M) && dF[x] != 0) continue;                 // H == F
                    if (j - 1 < pbi) continue;                                    // column j - 1 is not stored
                    int lit = lF[x];
                    if (lit == DIR_LIT_NONE) {
                        int kMl, uEl0, uEl1, dFl;
                        if (GAP == 1) { kMl = wl & 15; uEl0 = 0; uEl1 = 0; dFl = (wl >> DIRA_DF1_SH) & 7; }
                        else { kMl = wl & 15; uEl0 = (wl >> DIRC_UE1_SH) & 7; uEl1 = (wl >> DIRC_UE2_SH) & 31; dFl = x == 0 ? (wl >> DIRC_DF1_SH) & 7 : (wl >> DIRC_DF2_SH) & 31; }
                        const int h_is_hv = kMl != 0 || (GAP == 2 && (uEl0 == o1 || uEl1 == o2));
                        if (!h_is_hv && dFl > ox) { status = ABPOA_HIP_STATUS_NEED_SCORES; break; }      // the plane cannot decide this one (dir_plane.h)
                        lit = dir_f_origin(h_is_hv, dFl, ox);
                    }
                    if (lit == DIR_LIT_OPEN) { cur_op = OP_M | OP_E; hit = 1; }
                    else if (lit == DIR_LIT_EXT) { cur_op = bit; hit = 1; }
                }
                if (status != 0) break;
                if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (!hit && (cur_op & OP_M) && indel_first == 1) do_match(1);
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
            if (k_sel >= 0) {                                     // move to the chosen predecessor: its row record comes along (outside the window: the next turn stages a new one)
                i = __builtin_amdgcn_readlane(er.x, k_sel);
                cr = make_int4(__builtin_amdgcn_readlane(er.y, k_sel), __builtin_amdgcn_readlane(er.z, k_sel), __builtin_amdgcn_readlane(er2.x, k_sel), __builtin_amdgcn_readlane(er2.y, k_sel));
                cr2 = __builtin_amdgcn_readlane(er2.z, k_sel); cr_row = __builtin_amdgcn_readlane(er.w, k_sel) ? i : -1;
            }
        }
        walk_ticks = (long long)__builtin_amdgcn_s_memtime() - t_walk0;
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
        AlnOut o; for (int i_ = 0; i_ < 6; ++i_) o.seg[i_] = ts.seg[i_];
        o.status = status; o.best_score = best_score; o.best_row = best_i; o.best_col = best_j;
        o.node_s = node_s; o.node_e = node_e; o.query_s = query_s; o.query_e = query_e;
        o.n_aln_bases = n_aln; o.n_matched_bases = n_match; o.n_cigar = n_cigar; o.pad = -DB;      // pad < 0: direction-plane arena (bytes per word)
        o.n_cells = ts.n_cells; o.cells_used = ts.cursor;
        if (!(b.dbg & 128)) { o.seg[5] = win_ticks; o.seg[4] = (long long)n_windows * 1000; o.seg[3] = 0; o.seg[0] = 0; o.seg[1] = walk_ticks; o.seg[2] = (long long)bt_steps * 1000; }
        o.clk_dp = ts.clk1 - ts.clk0; o.clk_bt = (long long)__builtin_amdgcn_s_memtime() - ts.clk1; o.n_rows_done = ts.rows_done; o.n_bt_steps = bt_steps;
        *out_rec = o;
    }
}

}  // namespace abpoa_hip
