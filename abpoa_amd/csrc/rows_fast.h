#pragma once
#include "dp_common.h"
#include "dir_plane.h"
#include "rows_tight_asm.h"

namespace abpoa_hip {

// =====================================================================================================================
// Register-resident row loop ("fast loop") for the production case: global alignment, adaptive band, affine / convex
// gaps, every row active, band state at its reset value.  Same cells, bands and arg-max as the general loop in
// align_one() (and therefore as the reference), organised to minimise the INSTRUCTION COUNT of one row, because with
// one wavefront per SIMD a row costs (instructions x ~4 cycles) + exposed latency:
//   * max_pos_left/right are PULLED: left/right of row r = min/max over its predecessors p of (argmax_p + 1) (what the
//     reference's push at :1059-1067 leaves there once every predecessor is done), so the band needs no window at all;
//   * band geometry, arg-max and arena offset of the last 64 rows live in three VGPRs (lane = row & 63) and are read /
//     written with v_readlane / v_writelane -- no LDS round trip between a row and its successor;
//   * static per-row metadata (base, first four predecessors, remaining length) sits in VGPRs per 64-row tile
//     (lane = row & 63), loaded from the CSR arrays two tiles ahead;
//   * the H/E rows of the last fr_rows rows sit in an LDS ring as packed words (int16: H | E1 << 16), fr_cols columns per
//     row with "inf" guard cells on both sides and "inf" padding after the band: the first predecessor needs no range
//     masks at all (reading outside its stored band yields exactly what the reference reads or assigns there), further
//     predecessors need one unsigned compare per plane; one ds_read2_b32 fetches H[col-1], H|E[col];
//   * F is one 64-lane prefix-max scan: with g[c] = hs[c] + c*e,  F[c] = max_{c' < c} g[c'] - (oe - e) - c*e  whenever no
//     subtraction can wrap (same condition as fast_f_chain), then max with the lane-constant "inf injection" term of
//     the reference's zero-filled shifts;  vectors beyond max_pre_end_sn use the literal masked scan (set_f);
//   * rows that do not fit (predecessor further back than the ring, > 4 predecessors, band wider than the ring) take
//     the general gather (exact range masks, HBM copies) but share everything else.
// Arena format of the fast loop: one record of CW values per column -- {H, E1, F1, -} (affine) or {H, E1, E2, F1, F2, -, -, -}
// (convex), plane id = index in the record -- so that a row chunk is ONE wide store per lane instead of 3-5 two-byte ones
// (vector-memory instruction issue, not bytes, is what a lone wave pays for).
// (linear gaps, GAP == 0: {H, match flag})
template <typename T, int GAP> struct FastFmt { static constexpr int CW = GAP == 0 ? 2 : (GAP == 1 ? 4 : 8); };
// Direction-plane arenas (DIR = true, dir_plane.h): a row owns ONE word per column -- 2 bytes (affine) or 4 (convex) -- that records every
// decision the backtrack takes at that cell; only rows whose scores a later reader needs from HBM (a successor beyond the LDS score ring,
// the global best at the sink's predecessors, a row too wide for the ring) also keep their cell records, IN FRONT of the words (a row's arena
// offset is that of its words; the lanes past the band then write words that the next row overwrites, as the records always did).  Units of
// the arena cursor stay 32 bytes (one reference SIMD vector of scores); a row of nv vectors takes dir_units(nv) of them for its words.
template <typename T, int GAP> struct DirFmt {
    static constexpr int DB = GAP == 1 ? 2 : 4;                            // bytes per direction word
    static constexpr int UPV2 = Width<T>::PN * DB / 16;                    // 16-byte halves of a unit per vector of words: 2 / 1 (affine int16 / int32), 4 / 2 (convex)
    __device__ __host__ static constexpr int units(int nv) { return (nv * UPV2 + 1) >> 1; }
};
// a word from its predecessor fields kf = kM | kE1 << 4 (| kE2 << 8) and its uE / dF fields, nested so that every step is one shift-and-or (v_lshl_or_b32)
template <int GAP> __device__ __forceinline__ unsigned dir_word(unsigned kf, unsigned u1, unsigned u2, unsigned d1, unsigned d2) {
    static_assert(DIRA_DF1_SH - DIRA_UE1_SH == 3 && DIRC_UE2_SH - DIRC_UE1_SH == 3 && DIRC_DF2_SH - DIRC_DF1_SH == 3 && DIRC_DF1_SH - DIRC_UE1_SH == 8, "field layout of dir_plane.h");
    if (GAP == 1) return (((d1 << 3) | u1) << DIRA_UE1_SH) | kf;
    return (((((d2 << 3) | d1) << 8) | ((u2 << 3) | u1)) << DIRC_UE1_SH) | kf;
}
template <int GAP> __device__ __forceinline__ unsigned dir_pack(unsigned kM, unsigned kE1, unsigned kE2, unsigned u1, unsigned u2, unsigned d1, unsigned d2) {
    return dir_word<GAP>(GAP == 1 ? (kE1 << DIRA_KE1_SH) | kM : (((kE2 << 4) | kE1) << DIRC_KE1_SH) | kM, u1, u2, d1, d2);
}
__device__ __forceinline__ unsigned umin_(unsigned a, unsigned b) { return a < b ? a : b; }
// The reference's own comparisons for where F[j] came from (src/simd_abpoa_align.c:260-300), as the literal override of dir_plane.h; Hm1 / Fm1 =
// values of column j-1 of the same row
template <typename T> __device__ __forceinline__ unsigned dir_literal(int Hm1, int Fm1, int F, int oe, int e) {
    return (int)(T)(Hm1 - oe) == F ? (unsigned)DIR_LIT_OPEN : ((int)(T)(Fm1 - e) == F ? (unsigned)DIR_LIT_EXT : (unsigned)DIR_LIT_NEITHER);
}
template <typename T> struct FastIO {
    GLOBAL_AS const uint8_t *row_base, *row_sdist; GLOBAL_AS const int32_t *row_remain, *pred_off, *pred_row;
    GLOBAL_AS int32_t *g_bsn, *g_esn, *row_max_i, *g_left, *g_right; GLOBAL_AS int64_t *g_coff;
    T *planes;
};

// literal SIMD_SET_F for the vectors [nfast, ...) of one 64-lane chunk (global mode), reference :859-875 / :978-997
template <typename T, int GAP>
__device__ __forceinline__ void slow_f_vectors(int vbase, int end_sn, int max_pre, int nfast, int l, int vvl, T hs, T inf,
                                               T e1, T oe1, T o1, T e2, T oe2, T o2, T &F1, T &F2, T &first, T &first2) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
#pragma unroll
    for (int vv = 0; vv < NV; ++vv) {
        const int vg = vbase + vv;
        if (vv >= nfast && vg <= end_sn) {
            int set_num = PN;
            if (vg > max_pre) set_num = (vg == max_pre + 1) ? 2 : 1;
            T prev = (T)row_shr<1>((int)first, (int)hs);
            if (PN == 8) prev = (l == 0) ? first : prev;
            T f = wsub<T>(prev, oe1);
            f = set_f<T>(f, l, set_num, e1, inf);
            const T hlast = (T)__builtin_amdgcn_readlane((int)hs, vv * PN + PN - 1);
            first = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)f, vv * PN + PN - 1), o1));
            if (vvl == vv) F1 = f;
            if (GAP == 2) {
                T prev2 = (T)row_shr<1>((int)first2, (int)hs);
                if (PN == 8) prev2 = (l == 0) ? first2 : prev2;
                T g = wsub<T>(prev2, oe2);
                g = set_f<T>(g, l, set_num, e2, inf);
                first2 = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)g, vv * PN + PN - 1), o2));
                if (vvl == vv) F2 = g;
            }
        }
    }
}

#ifdef ABPOA_HIP_PROFILE
#define FSTAMP(I) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long t_ = (long long)__builtin_amdgcn_s_memtime(); fseg[I] += t_ - fseg_last; fseg_last = t_; }
#else
#define FSTAMP(I)
#endif
// Timing-only ablation switches (tools/kernel_bench.py with ABPOA_HIP_DBG=bits on the "prof" build): results are wrong on purpose.
// Diagnostic builds (-DABPOA_HIP_WIDE_COUNTERS, with ABPOA_HIP_DBG=128 so that the tail keeps them): rows per path of the wide loop in AlnOut.seg --
// 0 wide body, 1 not eligible (predecessor count / distance), 2 ring / geometry, 3 wider than NW chunks, 4 slow vectors span two wavefronts, 5 wrap guard
// -DABPOA_HIP_ROW_CENSUS (same hand-over): rows and clock ticks per body of the NARROW loop -- seg[i] = rows << 40 | ticks for i = 0 one predecessor
// (tight loop), 1 two predecessors (tight loop), 2 three / four predecessors (straight-line body), 3 the exact bodies (fast / general); 4 = tile switches
#ifdef ABPOA_HIP_ASM_CENSUS      // (-DABPOA_HIP_ROW_CENSUS -DABPOA_HIP_ASM_CENSUS: the census WITH the assembly loop -- slot 0 = its rows, slot 1 = the C++ copies of the one- / two-predecessor body)
#define CENSUS_SLOT(I) ((I) == 0 ? 1 : (I))
#else
#define CENSUS_SLOT(I) (I)
#endif
#ifdef ABPOA_HIP_ROW_CENSUS
#define CENSUS_T0() const long long cen_t0 = (long long)__builtin_amdgcn_s_memtime();
#define CENSUS(I) { fseg[CENSUS_SLOT(I)] += (1ll << 40) + ((long long)__builtin_amdgcn_s_memtime() - cen_t0); }
#else
#define CENSUS_T0()
#define CENSUS(I)
#endif
#ifdef ABPOA_HIP_WIDE_COUNTERS
#define WCOUNT(I) { fseg[I] += 1; }
#else
#define WCOUNT(I) {}
#endif
#ifdef ABPOA_HIP_ABLATE
#define ABL(BIT) (b.dbg & (BIT))
#else
#define ABL(BIT) false
#endif
template <typename T, int GAP, int NW = 1, bool WIDEB = false, bool DIR = false, bool XL = false>
__device__ __forceinline__ void rows_fast(const DevBatch &b, const AlnDesc &d, const FastIO<T> &io, const uint8_t *s_query,
                                          long long &cursor_out, long long &n_cells_out, int &status, int &rows_done_out, int &last_done, long long *fseg, int *best_out = nullptr) {
#ifdef ABPOA_HIP_PROFILE
    long long fseg_last = 0;
#endif
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int CW = FastFmt<T, GAP>::CW;          // values per arena cell record
    constexpr bool I16 = sizeof(T) == 2;
    // Records of the rows that keep their scores beside direction words in the WIDE kernel ("spill rows": read again from HBM by a far successor or by
    // the global best -- H and E only) are compact, CWR values per cell instead of CW: int16 {H, E1} / {H, E1, E2, -}, int32 affine {H, E1}, int32 convex
    // {H, (H - E1) | (H - E2) << 16} with 0xffff for "E is inf" (row 0).  8 bytes a cell instead of 16-32: with a 4-row ring almost half of the rows of a
    // 15 %-error graph are such rows, and the arenas of 2 048 x 10 kb convex read-sets would not fit otherwise.
    constexpr bool CSP = DIR && WIDEB;
    constexpr int CWR = CSP ? (I16 ? (GAP == 2 ? 4 : 2) : 2) : CW;
    constexpr bool CPK = CSP && !I16 && GAP == 2;
    static_assert(!(DIR && NW > 1), "teams of wavefronts keep the score-record arenas");
    static_assert(!(GAP == 0 && (DIR || NW > 1 || WIDEB)), "linear gaps: the narrow loop with H records only");
    constexpr int DB = DirFmt<T, GAP>::DB, CAPF1 = GAP == 1 ? DIRA_CAP1 : DIRC_CAP1, CAPF2 = DIRC_CAP2;
    auto dir_units = [](int nv) __attribute__((always_inline)) { return DirFmt<T, GAP>::units(nv); };
    // arena units of a row of nv vectors (spill: the row also keeps its score records)
    auto row_units = [&](int nv, bool spill) __attribute__((always_inline)) { return DIR ? dir_units(nv) + (spill ? nv * CWR : 0) : nv * CW; };
    bool row_spill = false;                          // (DIR) the current row keeps its score records: set by the row loop before a body runs
    constexpr bool WPLAN = NW > 1 || WIDEB;          // the wide kernels have their own score ring (LdsPlan wfr_*)
    // Ring words per column.  int16: H | E1 << 16, E2.  int32: H, E1, E2 -- but in the wide kernels' convex ring (EPACK) H and ONE word of differences
    // (H - E1) | (H - E2) << 16: E leaving a cell is max(Ein - e, H - oe) with Ein <= H, so H - E is in [e, oe] wherever H is a score, and 0 stands for
    // "both inf" where H is inf (outside the band, padding).  Two words instead of three: a ring of 8 rows for a 10 kb convex alignment is 29 KB, and
    // four workgroups share a CU (40 KB each) instead of three.  Row 0 -- E = inf beside real H -- stays out of such a ring: its successors read its records.
    constexpr bool EPACK = WPLAN && !I16 && GAP == 2;
    constexpr int NPW = I16 ? (GAP == 2 ? 2 : 1) : (GAP == 2 ? (EPACK ? 2 : 3) : (GAP == 0 ? 1 : 2));      // (linear gaps: H alone; int16: in the low half of the word)
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    constexpr int GEO_RING = 1 << 24;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int wid = NW > 1 ? __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : 0;      // wavefront of the workgroup = 64-column chunk of a wide row
    const int tid = NW > 1 ? (int)threadIdx.x : lane;
    constexpr int NT = NW * 64;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, m1 = b.m + 1, w = d.w;
    const int inf = d.inf_min;
    const int e1 = b.e1, o1 = b.o1, oe1 = b.o1 + b.e1, e2 = b.e2, o2 = b.o2, oe2 = b.o2 + b.e2;
    // (the single-wave wide kernel's ring width is a compile-time constant: address offsets and clamps fold into the instructions)
    const int RR = WPLAN ? b.lds.wfr_rows : b.lds.fr_rows, RC = WIDEB ? (XL ? WIDE_RING_COLS_XL : WIDE_RING_COLS) : (WPLAN ? b.lds.wfr_cols : b.lds.fr_cols), RCS = RC + 4;
    // (the wide kernels have their own carve-up of the LDS: the query packed two codes to a byte -- LdsPlan.w_*)
    const int ph_off = WPLAN ? b.lds.w_phase_off : b.lds.phase_off;
    int *fr = (int *)(lds_raw + ph_off + b.lds.fr_off);
    // wide rows: exchange slots (two parities x 8 entries of 16 bytes) and the hand-over record of a row done by wavefront 0 alone
    int4 *xch = (int4 *)(lds_raw + ph_off + b.lds.wx_off);
    int *bcast = (int *)(xch + 16);
    // LDS byte address of ring row (r & (RR - 1)), column 0, held by lane r & 63: RR divides 64, so one lane-constant VGPR serves
    // every row -- a v_readlane replaces the and / mul / shift / add chain per predecessor and for the row's own slot
    typedef __attribute__((address_space(3))) int lds_int_t;
    const int vslot = (int)(unsigned)(size_t)(lds_int_t *)fr + 4 * ((threadIdx.x & 63 & (RR - 1)) * (NPW * RCS) + 2);
    auto ring_at = [&](int slot_addr, int col_idx) __attribute__((always_inline)) { return (const int *)(lds_int_t *)(size_t)(unsigned)(slot_addr + 4 * col_idx); };
    int *s_mx = (int *)(lds_raw + (WPLAN ? b.lds.w_mx_off : b.lds.mx_off));
    // query code of base j (0-based): plain bytes, or -- wide kernels -- two 4-bit codes to a byte (ten thousand bases in 5 KB: with a 4-row ring an
    // alignment then needs under 20 KB of LDS, eight workgroups share a CU and every SIMD has two wavefronts to issue from)
    auto qat = [&](int j) __attribute__((always_inline)) -> int {
        if constexpr (WPLAN) return ((int)s_query[j >> 1] >> ((j & 1) * 4)) & 15; else return (int)s_query[j];
    };
    const int infw = I16 ? (int)(((unsigned)inf & 0xffffu) | ((unsigned)inf << 16)) : inf;
    const int qlen_sn = qlen / PN;
    auto wr = [](int x) __attribute__((always_inline)) { return (int)(T)x; };          // wrap to the score width

    // per-lane constants of the F scan and of the arg-max key
    const int idist = inj_dist<PN>(l);
    const int inj1 = idist >= 0 ? inf - idist * e1 : INT_MIN, inj2 = idist >= 0 ? inf - idist * e2 : INT_MIN;
    const int le1 = lane * e1, le2 = lane * e2;
    const int cf1 = oe1 - e1 + le1, cf2 = oe2 - e2 + le2;                               // F[c] = S[c] - cf
    const long long lo_ll = (long long)(I16 ? INT16_MIN : INT32_MIN) + imax(oe1, oe2) + (long long)PN * imax(e1, e2);
    const int fast_lo = (int)lo_ll;
    const int kconst = I16 ? (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (unsigned)(2047 - vvl)) : 0;
    const int ktie = ((PN - 1 - l) << 7) | (63 - vvl);                                  // int32 wide rows: residue, then (bit 6) the end vector, then the vector order

    // ---- LDS: extended score matrix (column m = 0) and the score ring, everything "inf"
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = tid; i < m * m1; i += NT) { const int bb = i / m1, qc = i - bb * m1; s_mx[i] = qc < m ? g_mat[bb * m + qc] : 0; } }
    for (int i = tid; i < RR * NPW * RCS; i += NT) { const int pl = (i / RCS) % NPW; fr[i] = (I16 && pl == 0) ? infw : ((EPACK && pl == 1) ? 0 : inf); }
    if (NW > 1 && tid < 16) xch[tid] = make_int4(INT_MIN, INT_MIN, I16 ? 0 : INT_MIN, 0);      // entries of absent wavefronts stay neutral
    WG_SYNC();
    auto ring_put = [&](int slot, int x, int H, int E1, int E2) __attribute__((always_inline)) {
        int *q = fr + slot * (NPW * RCS) + 2 + x;
        if (I16) { q[0] = (int)(((unsigned)H & 0xffffu) | ((unsigned)E1 << 16)); if (GAP == 2) q[RCS] = E2; }
        else { q[0] = H; if (GAP != 0) q[RCS] = E1; if (GAP == 2) q[2 * RCS] = E2; }
    };

    // extension mode (reference :1018-1026, set_extend_max_score): the rows are the global rows; after every row the running best cell (strictly greater: the
    // first row that reaches the maximum, in the reference's own row order) and, with z-drop, the test that ends the row loop
    const bool extend = sgpr(b.align_mode == ABPOA_HIP_EXTEND_MODE ? 1 : 0) != 0;
    int ex_best = inf, ex_i = 0, ex_j = 0, ex_rem = 0; bool zstop = false;
    const bool asm_tight_on = sgpr(!(b.dbg & 2048) && RC == 128 && b.align_mode != ABPOA_HIP_EXTEND_MODE ? 1 : 0) != 0;      // (rows_tight_asm.h; read once: the argument record lives in constant memory)
    int cur = 0, n_vec_lane = 0;                    // arena cursor in units of PN cells (one reference SIMD vector); cell count: per-lane sums of the flushed rows' vectors
    const int cap_pn = (int)(d.plane_cap / PN > 0x7fffffffLL ? 0x7fffffffLL : d.plane_cap / PN);
    const int cap_turbo = cap_pn - (DIR ? dir_units(NV) + NV * CWR : NV * CW);       // arena room test of the straight-line rows (at most NV vectors)
    const int remain_end = __builtin_amdgcn_readfirstlane(io.row_remain[gn - 1]);
    // ------------------------------------------------------------------ row 0, reference :553-662
    int vg_geo = 0, vg_mi = 0, vg_off = 0;          // lane = row & 63: beg_sn | end_sn << 12 | in-ring << 24, arg-max column, arena offset / PN
    int vg_vm = 0, rowmax = 0;                      // (wide kernel) the row's maximum H: the next rows centre their packed arg-max keys on it
    {
        const int r = __builtin_amdgcn_readfirstlane(io.row_remain[0]) - remain_end - 1;
        const int dp_end0 = imin(qlen, imax(0, qlen - r) + w);
        const int end_sn0 = dp_end0 / PN, W0 = (end_sn0 + 1) * PN;
        // (DIR: row 0 has no decisions to record; it keeps its score records because new branches anywhere in the graph start at the source)
        if ((long long)W0 * CWR > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; cursor_out = 0; n_cells_out = 0; rows_done_out = 0; return; }
        const bool ring0 = W0 <= RC && !EPACK;
        T *H = io.planes;
        for (int i = tid; i < W0; i += NT) {
            int h, x1 = inf, x2 = inf, f1 = inf, f2 = inf;
            if (GAP == 0) h = wr(-e1 * i);      // (reference :553-607 lg_first_row)
            else if (GAP == 1) { const int g = wr(-o1 - e1 * i); h = i == 0 ? 0 : g; x1 = i == 0 ? wr(-oe1) : inf; f1 = i == 0 ? inf : g; }
            else {
                const int g1 = wr(-o1 - e1 * i), g2 = wr(-o2 - e2 * i);
                h = i == 0 ? 0 : imax(g1, g2); x1 = i == 0 ? wr(-oe1) : inf; x2 = i == 0 ? wr(-oe2) : inf; f1 = i == 0 ? inf : g1; f2 = i == 0 ? inf : g2;
            }
            T *cellp = H + (long long)i * CWR;
            if constexpr (CPK) { cellp[0] = (T)h; cellp[1] = (T)(i == 0 ? (int)((unsigned)(h - x1) | ((unsigned)(h - x2) << 16)) : -1); }      // (E = inf beside a real H: 0xffff)
            else if constexpr (CSP) { cellp[0] = (T)h; cellp[PL_E1] = (T)x1; if (GAP == 2) { cellp[PL_E2] = (T)x2; cellp[3] = (T)0; } }
            else if constexpr (GAP == 0) { cellp[0] = (T)h; cellp[1] = (T)0; }
            else {
                cellp[0] = (T)h; cellp[PL_E1] = (T)x1; cellp[PL_F1] = (T)f1;
                if (GAP == 2) { cellp[PL_E2] = (T)x2; cellp[PL_F2] = (T)f2; }
            }
            if (ring0) ring_put(0, i, h, x1, x2);
        }
        cur = (end_sn0 + 1) * CWR;
        // (DIR: a row's offset is that of its -- here absent -- words, its records sit in front of them) // source: successors get left = right = 1 (:556-561)
        if (lane == 0) { vg_geo = (end_sn0 << 12) | (ring0 ? GEO_RING : 0); vg_mi = 0; vg_off = DIR ? cur : 0; }
        if (NW > 1) WG_SYNC();               // ring row 0 was written by every wavefront
    }

    // ------------------------------------------------------------------ static metadata, two tiles ahead
    struct MetaA { int ps, pe, base, rem, sd; };
    constexpr int NPM = 8;                          // predecessors kept in registers per row (wide rows: up to 8 take the common path)
    struct MetaB { int p[NPM]; };
    auto load_a = [&](int t0) __attribute__((always_inline)) {
        MetaA a; const int r = imin(t0 + lane, gn - 1);
        a.ps = io.pred_off[r]; a.pe = io.pred_off[r + 1]; a.base = io.row_base[r]; a.rem = io.row_remain[r]; a.sd = 0;
        if constexpr (DIR) a.sd = io.row_sdist[r];
        return a;
    };
    auto load_b = [&](const MetaA &a) __attribute__((always_inline)) {
        MetaB q; const int np = a.pe - a.ps;
#pragma unroll
        for (int k = 0; k < NPM; ++k) q.p[k] = io.pred_row[a.ps + imin(k, imax(np - 1, 0))];
        return q;
    };
    MetaA a1 = load_a(0); MetaB b1 = load_b(a1); MetaA a2 = load_a(64);
    int tv_r1 = 0, tv_r2 = 0;      // (assembly loop: the static terms of the band, rows_tight_asm.h TA_BAND_*)
    int tv_meta = 0, tv_rterm = 0, tv_ps = 0, tv_p0 = 0, tv_p1 = 0, tv_p2 = 0, tv_p3 = 0;
    int tv_p4 = 0, tv_p5 = 0, tv_p6 = 0, tv_p7 = 0;      // (wide rows only)
    int tv_tb = 0;          // turbo rows: dist(pred 0) | dist(pred 1) << 8 | (base * (m + 1) * 4) << 16
    auto switch_tile = [&](int t0) __attribute__((always_inline)) {
        const int myrow = t0 + lane, np = a1.pe - a1.ps;
        bool fastrow = np >= 1 && np <= 4 && myrow < gn - 1 && myrow >= 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int dk = myrow - b1.p[k]; fastrow = fastrow && dk >= 1 && dk < RR; }
        tv_meta = (a1.base & 0xff) | (imin(np, 255) << 8) | (fastrow ? (1 << 16) : 0);
        if (!WPLAN) {
            tv_meta |= ((fastrow && np <= 2 && RC <= 128) ? (1 << 17) : 0) | ((fastrow && np >= 3 && RC <= 128) ? (1 << 18) : 0);      // bit 17: straight-line body (pads 128 ring columns)
            bool row8 = np >= 5 && np <= 8 && myrow < gn - 1 && myrow >= 1 && RC <= 128;      // bit 20: five to eight predecessors, all in the score ring: the straight-line body's widest copy
#pragma unroll
            for (int k = 0; k < NPM; ++k) { const int dk = myrow - b1.p[k]; row8 = row8 && dk >= 1 && dk < RR; }
            tv_meta |= row8 ? (1 << 20) : 0;
        }
        {                                            // bit 19: the row may take the all-chunks body (1..8 predecessors, all inside the 64-row geometry ring)
            bool widerow = np >= 1 && np <= 8 && myrow < gn - 1 && myrow >= 1;
#pragma unroll
            for (int k = 0; k < NPM; ++k) { const int dk = myrow - b1.p[k]; widerow = widerow && dk >= 1 && dk < ((WIDEB || !WPLAN) ? 64 : RR); }      // (single-wave loops: older ones come from HBM)
            tv_meta |= widerow ? (1 << 19) : 0;
        }
        if constexpr (DIR) {      // bit 21: a successor beyond the score ring (or the sink) will read this row's H / E from HBM: it keeps its score records.
            // Such a row (2 % of them) leaves the tight loop -- whose copies of the straight-line body then carry no record code at all -- for the copy outside (bit 22)
            const bool sp = a1.sd >= RR;
            if (sp) { tv_meta |= 1 << 21; if (tv_meta & (1 << 17)) tv_meta = (tv_meta & ~(1 << 17)) | (1 << 22); }
        }
        tv_tb = ((myrow - b1.p[0]) & 0xff) | (((myrow - b1.p[1]) & 0xff) << 8) | (((a1.base & 0xff) * m1 * 4) << 16);
        tv_rterm = qlen - (a1.rem - remain_end - 1); tv_ps = a1.ps;
        tv_r1 = imin(gn, tv_rterm) - w; tv_r2 = tv_rterm + w;
        tv_p0 = b1.p[0]; tv_p1 = b1.p[1]; tv_p2 = b1.p[2]; tv_p3 = b1.p[3];
        tv_p4 = b1.p[4]; tv_p5 = b1.p[5]; tv_p6 = b1.p[6]; tv_p7 = b1.p[7];
        a1 = a2; b1 = load_b(a1); a2 = load_a(t0 + 128);
    };
    int qc_beg_sn = -1, qoff0 = 0, qoff1 = 0;        // cached query code of this lane's column for chunks 0/1 of band start qc_beg_sn
    auto geo_of = [&](int p, int row, int &geo, int &mi, int &off) __attribute__((always_inline)) {
        if (row - p < 64) { const int sl = p & 63; geo = __builtin_amdgcn_readlane(vg_geo, sl); mi = __builtin_amdgcn_readlane(vg_mi, sl); off = __builtin_amdgcn_readlane(vg_off, sl); }
        else {
            geo = __builtin_amdgcn_readfirstlane(gld_i32(io.g_bsn + p) | (gld_i32(io.g_esn + p) << 12)); mi = __builtin_amdgcn_readfirstlane(gld_i32(io.row_max_i + p));
            off = __builtin_amdgcn_readfirstlane((int)(uint32_t)(gld_i64(io.g_coff + p) / PN));
        }
    };
    // ---- per-row working set shared by the two row bodies and the epilogue
    int beg_sn = 0, end_sn = 0, off_pn = 0, max_pe = 0, rterm = 0, base = 0, np = 0;
    bool to_ring = false;
    unsigned am_key = 0; int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false;

    // band of the row from (min, max) predecessor arg-max and predecessor geometry, reference :710-720
    auto set_band = [&](auto pin, int mn_mi, int mx_mi, int min_pb) __attribute__((always_inline)) {
        auto S = [](int x) __attribute__((always_inline)) { if constexpr (decltype(pin)::value) return sgpr(x); else return x; };
        const int left = S(imin(gn, mn_mi + 1)), right = S(decltype(pin)::value ? mx_mi + 1 : imax(0, mx_mi + 1));      // (row arg-max >= -1)
        const int lo = S(imin(left, rterm) - w), hi = S(imax(right, rterm) + w);
        const int beg = S(imax(0, lo)), end = S(imin(qlen, hi));
        beg_sn = imax((int)((unsigned)beg / PN), min_pb); end_sn = (int)((unsigned)end / PN);
    };
    auto refresh_qc = [&]() __attribute__((always_inline)) {
        if (beg_sn != qc_beg_sn) {                 // band start moved: refresh this lane's cached query codes
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? qat(c0 - 1) : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? qat(c1 - 1) : m;
        }
    };
    // one predecessor's contribution from the score ring (k == 0: unmasked, see the header comment)
    // kb: 1 + list index of the first predecessor that supplies the maximum of H[.][col-1] (kidx = this one's 1 + list index): the match flag
    // kE1 / kE2 (DIR): 1 + list index of the first predecessor that holds the maximum of E1 / E2 entering the cell (dir_plane.h)
    auto from_ring = [&](int k, int p, int g_, int col, int &Mv, int &E1v, int &E2v, int &kb, int kidx, int &kE1, int &kE2, int q = 0) __attribute__((always_inline)) {
        const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
        const int x = col - pb * PN;
        const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), med3i(x - 1, -2, RC));
        if constexpr (GAP == 0) {      // linear gaps (reference :722-761): max(H[col-1] + q, H[col] - e) over the predecessor's vectors [max(pb, beg_sn), min(pe + 1, end_sn)], "inf" outside -- the first one too
            const int w0 = src[0], w1 = src[1];
            const int hm1 = I16 ? (int)(short)w0 : w0, h0 = I16 ? (int)(short)w1 : w1;
            const int t = imax(wr(hm1 + q), wr(h0 - e1));
            const bool inH = (unsigned)x < (unsigned)(Wp + PN);
            Mv = k == 0 ? (inH ? t : inf) : (inH ? imax(Mv, t) : Mv);
            return;
        }
        int hm1, ev1, ev2 = inf;
        if (I16) { const int w0 = src[0], w1 = src[1]; hm1 = (int)(short)w0; ev1 = w1 >> 16; if (GAP == 2) ev2 = src[RCS + 1]; }
        else if (EPACK) { hm1 = src[0]; const int h0 = src[1]; const unsigned dd = (unsigned)src[RCS + 1]; ev1 = h0 - (int)(dd & 0xffffu); ev2 = h0 - (int)(dd >> 16); }
        else { hm1 = src[0]; ev1 = src[RCS + 1]; if (GAP == 2) ev2 = src[2 * RCS + 1]; }
        if (k == 0) { Mv = hm1; E1v = ev1; E2v = ev2; kb = kidx; kE1 = kidx; kE2 = kidx; }
        else {
            const bool inH = (unsigned)x < (unsigned)(Wp + PN), inE = (unsigned)x < (unsigned)Wp;
            kb = (inH && hm1 > Mv) ? kidx : kb;
            if (DIR) { kE1 = (inE && ev1 > E1v) ? kidx : kE1; if (GAP == 2) kE2 = (inE && ev2 > E2v) ? kidx : kE2; }
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        }
    };
    // everything of a chunk after the predecessor gather: F, H, E, stores, ring, arg-max candidate (reference :854-883 / :972-1008)
    int ct_pH = 0, ct_pF1 = 0, ct_pF2 = 0;          // (DIR) H / F1 / F2 of the previous chunk's last column: the left neighbour of this chunk's first one
    auto chunk_tail = [&](int c, int nch, int Wr, int Mv, int E1v, int E2v, int q, int kb, int &first, int &first2, T *H, int my_slot, int kE1 = 1, int kE2 = 1) __attribute__((always_inline)) {
        const int rel = c * 64 + lane, col = beg_sn * PN + rel, vb = beg_sn + c * NV, v = vb + vvl;
        const bool in_band = rel < Wr;
        if constexpr (GAP == 0) {
            // linear gaps (reference :762-778): Mv already is max over the predecessors of max(H[col-1] + q, H[col] - e); the in-row term runs on H itself.  Vectors that use
            // the plain scan (up to max_pre_end_sn), while nothing can wrap: H[c] = max(prefix max of (h + c e) - c e, inf) over the whole chunk with the carry
            // entering at lane 0 (the clamp at inf is the reference's: `first` holds inf in its lanes 1 .. pn - 1; tests/test_linear_closed_form.py); the vectors beyond (set_num 1 / 0) and everything near the wrap limit: the literal masked scan
            int hl = Mv;
            if (c == 0) first = __builtin_amdgcn_readlane(hl, 0);
            const int nvec = imin(NV, end_sn - vb + 1);
            int nfast = imax(0, imin(nvec, max_pe - vb + 1));
            if (nfast > 0 && (first < fast_lo || __any(vvl < nfast && hl < fast_lo))) nfast = 0;
            if (nfast > 0) {
                int g = hl + le1; g = lane == 0 ? imax(g, first) : g;
                const int Hc = imax(wave_scan_max_i32(g) - le1, inf);
                hl = vvl < nfast ? Hc : hl;
                first = __builtin_amdgcn_readlane(Hc, nfast * PN - 1) - e1;
            }
#pragma unroll
            for (int vv = 0; vv < NV; ++vv) {
                if (vv >= nfast && vb + vv <= end_sn) {
                    const int vg = vb + vv;
                    const int set_num = vg > max_pe ? (vg == max_pe + 1 ? 1 : 0) : PN;
                    T hv = tmax<T>((T)hl, l == 0 ? (T)first : (T)inf);
                    hv = set_f<T>(hv, l, set_num, (T)e1, (T)inf);
                    if (vvl == vv) hl = (int)hv;
                    first = (int)wsub<T>((T)__builtin_amdgcn_readlane((int)hv, vv * PN + PN - 1), (T)e1);
                }
            }
            const int Hout = hl;
            if (!ABL(1)) { if (I16) *(int *)(H + (long long)rel * CW) = Hout & 0xffff; else { int2 rec; rec.x = Hout; rec.y = 0; *(int2 *)(H + (long long)rel * CW) = rec; } }      // ({H, flag 0 = not known}; lanes past the band write cells the next row overwrites)
            if (to_ring && !ABL(2)) { int *qd = fr + my_slot + 2 + rel; qd[0] = in_band ? (I16 ? (int)(((unsigned)Hout & 0xffffu) | ((unsigned)inf << 16)) : Hout) : infw; }
            if (!ABL(4)) {
                const bool is_end = (v == end_sn);
                int cand = Hout;
                if (end_sn == qlen_sn) cand = (is_end && col > qlen) ? inf : cand;
                if (I16) {
                    const unsigned key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
                    am_key = (in_band && key > am_key) ? key : am_key;
                } else if (in_band && (!am_any || (is_end ? cand >= am_val : cand > am_val))) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
            }
            return;
        }
        const int h = wr(Mv + q);
        int hs = h; if (GAP == 2) hs = imax(imax(h, E1v), E2v);
        if (c == 0) { first = __builtin_amdgcn_readlane(h, 0); first2 = first; }
        const int nvec = imin(NV, end_sn - vb + 1);
        int nfast = imin(nvec, max_pe - vb + 1);
        if (nfast < 0) nfast = 0;
        if (nfast > 0 && __any(vvl < nfast && h < fast_lo)) nfast = 0;
        int F1 = inf, F2 = inf;
        if (nfast > 0 && !ABL(8)) {
            const int g1 = hs + le1;
            const int S1 = wave_scan_max_i32(wave_shr1(first - e1, g1));
            F1 = imax(S1 - cf1, inj1);
            if (GAP == 2) { const int g2 = hs + le2; const int S2 = wave_scan_max_i32(wave_shr1(first2 - e2, g2)); F2 = imax(S2 - cf2, inj2);
                            if (nfast < nvec || c + 1 < nch) { const int lastl = nfast * PN - 1; first2 = __builtin_amdgcn_readlane(imax(S2, g2), lastl) - lastl * e2; } }
            if (nfast < nvec || c + 1 < nch) { const int lastl = nfast * PN - 1; first = __builtin_amdgcn_readlane(imax(S1, g1), lastl) - lastl * e1; }
        }
        if (nfast < nvec && !ABL(32)) {
            T f1t = (T)F1, f2t = (T)F2, fi = (T)first, fi2 = (T)first2;
            slow_f_vectors<T, GAP>(vb, end_sn, max_pe, nfast, l, vvl, (T)hs, (T)inf, (T)e1, (T)oe1, (T)o1, (T)e2, (T)oe2, (T)o2, f1t, f2t, fi, fi2);
            F1 = (int)f1t; F2 = (int)f2t; first = (int)fi; first2 = (int)fi2;
        }
        FSTAMP(2)
        int Hout, E1out, E2out = inf;
        if (GAP == 1) {
            const int tmp = imax(h, E1v);
            Hout = imax(tmp, F1);
            const int en = imax(wr(E1v - e1), wr(Hout - oe1));
            E1out = (Hout == tmp) ? en : inf;
        } else {
            Hout = imax(hs, imax(F1, F2));
            E1out = imax(wr(E1v - e1), wr(Hout - oe1));
            E2out = imax(wr(E2v - e2), wr(Hout - oe2));
        }
        // one record store per lane, all 64 lanes (lanes past the band write into cells the NEXT row overwrites: same wave,
        // program order; the arena carries 64 records of slack at its end)
        const int he = (int)(((unsigned)Hout & 0xffffu) | ((unsigned)E1out << 16));      // int16: also the score-ring word
        // match flag for the backtrack (see turbo_body): compared without wrapping, as the reference's backtrack does (:130-160)
        const int mflag = (Mv + q == Hout && kb <= 64) ? kb : 0;
        T *Hrec = H;                                  // where the row's score records go
        if constexpr (DIR) {
            // the exact bodies record every comparison literally (dir_plane.h): H == Ein, E opened from H, H == F, and -- as the override -- where F came
            // from, with the reference's own tests on the stored neighbours (:260-300); the left neighbour of the chunk's first column is carried over
            Hrec = H - (long long)(end_sn - beg_sn + 1) * CWR * PN;      // (a row that keeps its records has them in front of its words)
            const int Hm1 = wave_shr1(ct_pH, Hout), F1m1 = wave_shr1(ct_pF1, F1);
            const int en1_ = GAP == 1 ? imax(wr(E1v - e1), wr(Hout - oe1)) : E1out;      // (affine: E's maximum before the reference replaces it by "inf" where H is an F term, :880)
            const unsigned u1 = Hout == E1v ? (unsigned)o1 : (en1_ == wr(Hout - oe1) ? 0u : (o1 > 1 ? 1u : 0u));
            const unsigned d1 = Hout == F1 ? 0u : (unsigned)CAPF1;
            unsigned l1 = rel >= 1 ? dir_literal<T>(Hm1, F1m1, F1, oe1, e1) : 0u, l2 = 0, u2 = 0, d2 = 0;
            if (GAP == 2) {
                const int F2m1 = wave_shr1(ct_pF2, F2);
                u2 = Hout == E2v ? (unsigned)o2 : (E2out == wr(Hout - oe2) ? 0u : (o2 > 1 ? 1u : 0u));
                d2 = Hout == F2 ? 0u : (unsigned)CAPF2;
                l2 = rel >= 1 ? dir_literal<T>(Hm1, F2m1, F2, oe2, e2) : 0u;
            }
            const unsigned wd = dir_pack<GAP>((unsigned)mflag, (unsigned)kE1, (unsigned)kE2, u1, u2, d1, d2) | (GAP == 1 ? l1 << DIRA_LF1_SH : (l1 << DIRC_LF1_SH | l2 << DIRC_LF2_SH));
            // (lanes past the band write words the next row overwrites: same wave, program order)
            if (GAP == 1) *(uint16_t *)((char *)H + rel * DB) = (uint16_t)wd; else *(uint32_t *)((char *)H + rel * DB) = wd;
            ct_pH = __builtin_amdgcn_readlane(Hout, 63); ct_pF1 = __builtin_amdgcn_readlane(F1, 63); if (GAP == 2) ct_pF2 = __builtin_amdgcn_readlane(F2, 63);
        }
        // (DIR: only a row that keeps its records stores them, and only its in-band lanes do: the row's words start right behind its last record)
        if (ABL(1) || (DIR && !(row_spill && in_band))) {}
        else if (CSP) {      // compact records (CWR): what a far successor and the global best read
            if (I16 && GAP == 1) *(int *)(Hrec + (long long)rel * CWR) = he;
            else if (I16) { int2 rec; rec.x = he; rec.y = E2out & 0xffff; *(int2 *)(Hrec + (long long)rel * CWR) = rec; }
            else if (GAP == 1) { int2 rec; rec.x = Hout; rec.y = E1out; *(int2 *)(Hrec + (long long)rel * CWR) = rec; }
            else { int2 rec; rec.x = Hout; rec.y = (int)((unsigned)(Hout - E1out) | ((unsigned)(Hout - E2out) << 16)); *(int2 *)(Hrec + (long long)rel * CWR) = rec; }
        }
        else if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(Hrec + (long long)rel * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2 & 0xffff; rec.w = mflag; *(int4 *)(Hrec + (long long)rel * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(Hrec + (long long)rel * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(Hrec + (long long)rel * CW); dst[0] = r0;
                dst[1] = r1; }
        if (to_ring && !ABL(2)) {
            int *qd = fr + my_slot + 2 + rel;
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
            else if (EPACK) { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? (int)((unsigned)(Hout - E1out) | ((unsigned)(Hout - E2out) << 16)) : 0; }
            else { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS] = in_band ? E2out : inf; }
        }
        if (!ABL(4)) {   // running arg-max candidate of this lane, reference :1043-1057
            const bool is_end = (v == end_sn);
            int cand = Hout;
            if (end_sn == qlen_sn) cand = (is_end && col > qlen) ? inf : cand;
            if (I16) {
                const unsigned key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
                am_key = (in_band && key > am_key) ? key : am_key;
            } else if (in_band && (!am_any || (is_end ? cand >= am_val : cand > am_val))) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
        }
    };
    auto pad_ring = [&](int nch, int my_slot) __attribute__((always_inline)) {        // "inf" after the band, up to the ring width
        if (!ABL(2)) for (int c = nch; c < (RC >> 6); ++c) {
            int *qd = fr + my_slot + 2 + c * 64 + lane;
            qd[0] = infw; if (NPW > 1) qd[RCS] = EPACK ? 0 : inf; if (NPW > 2) qd[2 * RCS] = inf;
        }
    };
    // reserve the row's arena cells; false = overflow
    auto reserve = [&]() __attribute__((always_inline)) {
        const int nvr = end_sn - beg_sn + 1;
        if (cur + row_units(nvr, row_spill) > cap_pn) return false;
        off_pn = cur + ((DIR && row_spill) ? nvr * CWR : 0); cur += row_units(nvr, row_spill);
        return true;
    };


    // ---- TURBO body: the dominant row shape as straight-line code -- 1 or 2 predecessors (both in the rings), band of at
    //      most 64 columns (one chunk), every vector takes the closed-form F scan (end_sn <= max_pre_end_sn), not the last
    //      query vector, and no value near the wrap limit.  Exactly two rarely-taken exits, both before any side effect.
    //      Returns 1 = done (mi set), 0 = not applicable.
    int mi = -1;
    const int lane4 = lane * 4;
    // arg-max key constants (normal / end_sn vector): lane residue, vector priority, and -- never decisive, it only saves the decoding -- the lane
    const int kN = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)(NV - 1 - vvl) << 8) | (unsigned)lane),
            kE = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (8u << 8) | (unsigned)lane);
    // ---- the straight-line row for LINEAR gaps (reference simd_abpoa_lg_dp :701-779; same interface and exits as turbo_body below): per predecessor ONE
    //      ds_read2 (H[col-1], H[col]), max(H[col-1] + q, H[col] - e) under the reference's vector range, the in-row term as one 64-lane prefix-max scan on H itself
    //      with the arg-max key riding beside it (e >= 1: a propagated term never holds the row maximum), one H store per lane.
    auto turbo_lin = [&](auto npc, auto slowc, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr bool SLOWV = decltype(slowc)::value;
        constexpr int NPC = decltype(npc)::value, NPL = NPC == 1 ? 1 : (NPC == 2 ? 2 : (NPC == 4 ? 4 : 8));
        const int tb = __builtin_amdgcn_readlane(tv_tb, ti);
        int px[NPL], gx[NPL];
        px[0] = row - (tb & 0xff);
        if constexpr (NPC >= 2) px[1] = row - ((tb >> 8) & 0xff);
        if constexpr (NPC >= 4) { px[2] = __builtin_amdgcn_readlane(tv_p2, ti); px[3] = np > 3 ? __builtin_amdgcn_readlane(tv_p3, ti) : px[2]; }      // (a missing fourth repeats the third)
        if constexpr (NPC == 8) { px[4] = __builtin_amdgcn_readlane(tv_p4, ti); px[5] = __builtin_amdgcn_readlane(tv_p5, ti); px[6] = __builtin_amdgcn_readlane(tv_p6, ti); px[7] = __builtin_amdgcn_readlane(tv_p7, ti); }
        int mn = 0, mx = 0, min_pb = 0, ring = -1; max_pe = 0;
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            gx[k] = __builtin_amdgcn_readlane(vg_geo, px[k] & 63); const int mk = __builtin_amdgcn_readlane(vg_mi, px[k] & 63);
            if (k == 0) { mn = mx = mk; min_pb = gx[0] & 0xfff; max_pe = (gx[0] >> 12) & 0xfff; }
            else { mn = imin(mn, mk); mx = imax(mx, mk); min_pb = imin(min_pb, gx[k] & 0xfff); max_pe = imax(max_pe, (gx[k] >> 12) & 0xfff); }
            ring &= gx[k];
        }
        mn = sgpr(mn); mx = sgpr(mx);
        set_band(std::true_type{}, mn, mx, min_pb);
        const int nvr = end_sn - beg_sn + 1;
        const int okbits = (nvr - NV - 1) & ((SLOWV ? beg_sn : end_sn) - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7);      // (as turbo_body)
        if (__builtin_expect(okbits >= 0, 0)) return (!SLOWV && ((nvr - NV - 1) & (beg_sn - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7)) < 0) ? 0 : -1;
        const int Wr = nvr * PN;
        if (__builtin_expect(beg_sn != qc_beg_sn, 0)) {
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? qat(c0 - 1) : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? qat(c1 - 1) : m;
        }
        const int q = *(const int *)((const char *)s_mx + (tb >> 16) + qoff0 * 4);
        const int colrel = beg_sn * PN + lane;
        int ra[NPL], rb[NPL], xk[NPL], Wk[NPL];
#pragma unroll
        for (int k = 0; k < NPL; ++k) {      // every predecessor's words in flight together
            const int pbk = gx[k] & 0xfff; Wk[k] = (((gx[k] >> 12) & 0xfff) - pbk + 1) * PN; xk[k] = colrel - pbk * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, px[k]), med3i(xk[k] - 1, -2, RC));
            ra[k] = src[0]; rb[k] = src[1];
        }
        const bool in_band = lane < Wr;
        const int key_c = (vvl == nvr - 1) ? kE : kN;
        const int qd_addr = __builtin_amdgcn_readlane(vslot, ti) + 4 * lane;
        const unsigned rec_off = (unsigned)(cur * (int)(PN * sizeof(T)) + lane * (int)(CW * sizeof(T)));
        asm volatile("" :: "v"(key_c), "v"(qd_addr), "v"(rec_off));
        __builtin_amdgcn_sched_barrier(0);
        int h = inf, dmax = INT_MIN, kfirst = 0;      // (dmax / kfirst: the largest diagonal term and 1 + list index of the first predecessor that holds it -- the record's match flag)
#pragma unroll
        for (int k = 0; k < NPL; ++k) {
            asm volatile("" : "+v"(ra[k]), "+v"(rb[k]));
            const int hm1 = I16 ? (int)(short)ra[k] : ra[k], h0 = I16 ? (int)(short)rb[k] : rb[k];
            const int dg = hm1 + q, t = imax(dg, h0 - e1);         // (no wrap: checked below, and "inf" leaves room for q and e)
            const bool inH = (unsigned)xk[k] < (unsigned)(Wk[k] + PN);
            h = k == 0 ? (inH ? t : inf) : (inH ? imax(h, t) : h);
            if (k == 0) { dmax = inH ? dg : INT_MIN; kfirst = 1; } else { kfirst = (inH && dg > dmax) ? k + 1 : kfirst; dmax = inH ? imax(dmax, dg) : dmax; }
        }
        const bool near_wrap = __any(in_band && h < fast_lo);
        const bool am_ok = in_band && colrel <= qlen;
        unsigned akey = I16 ? (am_ok ? ((unsigned)h << 16) + (unsigned)key_c : 0u) : 0u;
        int aval = am_ok ? h : INT_MIN;
        int s_ = h + le1;                                          // (lane 0: its own value is the reference's `first`)
        if (I16) wave_scan2_iu(s_, akey); else wave_scan2_ii(s_, aval);
        int Hout = imax(s_ - le1, inf);                             // (the reference clamps every lane at inf: `first` holds inf in its lanes 1 .. pn - 1; tests/test_linear_closed_form.py)
        if (__builtin_expect(near_wrap, 0)) return -1;
        if (SLOWV && end_sn > max_pe) {                             // vectors beyond every predecessor's band: the literal masked scan on H (set_num 1 / 0, reference :766-772)
            const int nfast = max_pe - beg_sn + 1;                  // 1 <= nfast < nvr
            int first = __builtin_amdgcn_readlane(Hout, nfast * PN - 1) - e1, hl = vvl < nfast ? Hout : h;
#pragma unroll
            for (int vv = 1; vv < NV; ++vv) {
                if (vv >= nfast && vv < nvr) {
                    const int set_num = vv == nfast ? 1 : 0;
                    T hv = tmax<T>((T)hl, l == 0 ? (T)first : (T)inf);
                    hv = set_f<T>(hv, l, set_num, (T)e1, (T)inf);
                    if (vvl == vv) hl = (int)hv;
                    first = (int)wsub<T>((T)__builtin_amdgcn_readlane((int)hv, vv * PN + PN - 1), (T)e1);
                }
            }
            Hout = hl;
        }
        // ---- from here on the row is committed
        off_pn = cur; cur += row_units(nvr, false);
        // {H, 1 + index of the first predecessor k (list order) with H[k][col-1] + q == H[col], 0 = none}: the backtrack's match runs (backtrack.h PL_FLAG)
        { const int mflag = dmax == Hout ? kfirst : 0; char *const dp = (char *)io.planes + (size_t)rec_off;
          if (I16) *(int *)dp = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)Hout, 0x05040100u); else { int2 rec; rec.x = Hout; rec.y = mflag; *(int2 *)dp = rec; } }
        {
            int *qd = (int *)ring_at(qd_addr, 0);
            qd[0] = in_band ? (I16 ? (int)__builtin_amdgcn_perm((unsigned)inf, (unsigned)Hout, 0x05040100u) : Hout) : infw;
            qd[64] = infw;                                          // (RC <= 128 for these rows: tv_meta bit 17)
        }
        if (I16) {
            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)akey, 63);
            rowmax = (int)(kb >> 16) - 32768;
            mi = (rowmax > inf) ? beg_sn * PN + (int)(kb & 63) : -1;
        } else {
            const int vmax = __builtin_amdgcn_readlane(aval, 63);
            rowmax = vmax;
            const unsigned key = (am_ok && h == vmax) ? (((unsigned)(PN - 1 - l) << 12) | (unsigned)((vvl == nvr - 1) ? 8 : NV - 1 - vvl)) : 0u;
            const unsigned kb = wave_max_u32_s(key);
            const int vrel = (kb & 8) ? nvr - 1 : NV - 1 - (int)(kb & 7);
            mi = (vmax > inf) ? (beg_sn + vrel) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)) : -1;
        }
        return 1;
    };
    auto turbo_aff = [&](auto npc, auto slowc, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr bool SLOWV = decltype(slowc)::value;      // handles vectors beyond every predecessor's band (the tight loop's copies do not: they decline and the row comes back here)
        constexpr int NPC = decltype(npc)::value;
        const int tb = __builtin_amdgcn_readlane(tv_tb, ti);
        const int p0 = row - (tb & 0xff);
        const int g0 = __builtin_amdgcn_readlane(vg_geo, p0 & 63), m0 = __builtin_amdgcn_readlane(vg_mi, p0 & 63);
        const int pb0 = g0 & 0xfff, pe0 = (g0 >> 12) & 0xfff;
        int mn = m0, mx = m0, min_pb = pb0, ring = g0; max_pe = pe0;
        int p1 = p0, g1 = g0, p2 = p0, g2 = g0, p3 = p0, g3 = g0;
        if (NPC >= 2) {
            p1 = row - ((tb >> 8) & 0xff);
            g1 = __builtin_amdgcn_readlane(vg_geo, p1 & 63); const int m1_ = __builtin_amdgcn_readlane(vg_mi, p1 & 63);
            mn = sgpr(imin(m0, m1_)); mx = sgpr(imax(m0, m1_)); min_pb = imin(pb0, g1 & 0xfff); max_pe = imax(pe0, (g1 >> 12) & 0xfff); ring &= g1;
        }
        if (NPC >= 4) {      // three or four predecessors (np at run time); a missing fourth repeats the third
            p2 = __builtin_amdgcn_readlane(tv_p2, ti); p3 = np > 3 ? __builtin_amdgcn_readlane(tv_p3, ti) : p2;
            g2 = __builtin_amdgcn_readlane(vg_geo, p2 & 63); g3 = __builtin_amdgcn_readlane(vg_geo, p3 & 63);
            const int m2_ = __builtin_amdgcn_readlane(vg_mi, p2 & 63), m3_ = __builtin_amdgcn_readlane(vg_mi, p3 & 63);
            mn = sgpr(imin(mn, imin(m2_, m3_))); mx = sgpr(imax(mx, imax(m2_, m3_)));
            min_pb = imin(min_pb, imin(g2 & 0xfff, g3 & 0xfff)); max_pe = imax(max_pe, imax((g2 >> 12) & 0xfff, (g3 >> 12) & 0xfff)); ring &= g2 & g3;
        }
        int px[4] = {p0, p0, p0, p0}, gx[4] = {g0, g0, g0, g0};
        if (NPC == 8) {      // five to eight predecessors: the list entries past the last one repeat it (load_b), so all eight are read
            px[0] = __builtin_amdgcn_readlane(tv_p4, ti); px[1] = __builtin_amdgcn_readlane(tv_p5, ti); px[2] = __builtin_amdgcn_readlane(tv_p6, ti); px[3] = __builtin_amdgcn_readlane(tv_p7, ti);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                gx[j] = __builtin_amdgcn_readlane(vg_geo, px[j] & 63); const int mj_ = __builtin_amdgcn_readlane(vg_mi, px[j] & 63);
                mn = imin(mn, mj_); mx = imax(mx, mj_); min_pb = imin(min_pb, gx[j] & 0xfff); max_pe = imax(max_pe, (gx[j] >> 12) & 0xfff); ring &= gx[j];
            }
            mn = sgpr(mn); mx = sgpr(mx);
        }
        set_band(std::true_type{}, mn, mx, min_pb);
        const int nvr = end_sn - beg_sn + 1;
        // all conditions as sign bits: (x <= y) <=> (x - y - 1) < 0
        // (arena room: checked for a full-width row, cap_turbo = cap_pn - NV * CW)
        // (vectors beyond every predecessor's band -- one new vector every PN rows as the band moves right -- are taken below with the literal
        //  masked scan; the closed form needs at least the first vector inside: max_pe >= beg_sn)
        const int okbits = (nvr - NV - 1) & ((SLOWV ? beg_sn : end_sn) - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7);      // GEO_RING (bit 24) -> bit 31
        // (declined: 0 when the copy that takes vectors beyond the predecessors' bands would accept the row, else -1 -- the caller then goes straight to the exact bodies)
        if (__builtin_expect(okbits >= 0, 0)) return (!SLOWV && ((nvr - NV - 1) & (beg_sn - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7)) < 0) ? 0 : -1;
        const int Wr = nvr * PN;
        if (__builtin_expect(beg_sn != qc_beg_sn, 0)) {
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? qat(c0 - 1) : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? qat(c1 - 1) : m;
        }
        const int q = *(const int *)((const char *)s_mx + (tb >> 16) + qoff0 * 4);
        const int colrel = beg_sn * PN + lane;                     // this lane's column
        int Mv, E1v, E2v = inf, raw0, raw1, raw2 = inf;            // the first predecessor's words as loaded (decoded after the block below)
        {
            const int x = colrel - pb0 * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p0), med3i(x - 1, -2, RC));
            if (I16) { raw0 = src[0]; raw1 = src[1]; if (GAP == 2) raw2 = src[RCS + 1]; }
            else { raw0 = src[0]; raw1 = src[RCS + 1]; if (GAP == 2) raw2 = src[2 * RCS + 1]; }
        }
        // the second predecessor's words go out with the first one's: both LDS reads are in flight together
        int rb0 = 0, rb1 = 0, rb2 = inf, x1 = 0, Wp1 = 0;
        int rc0 = 0, rc1 = 0, rc2 = inf, x2 = 0, Wp2 = 0, rd0 = 0, rd1 = 0, rd2 = inf, x3 = 0, Wp3 = 0;
        int re0[4] = {0, 0, 0, 0}, re1[4] = {0, 0, 0, 0}, re2[4] = {inf, inf, inf, inf}, xe[4] = {0, 0, 0, 0}, We[4] = {0, 0, 0, 0};
        if (NPC == 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int pbj = gx[j] & 0xfff; We[j] = (((gx[j] >> 12) & 0xfff) - pbj + 1) * PN; xe[j] = colrel - pbj * PN;
                const int *sj = ring_at(__builtin_amdgcn_readlane(vslot, px[j]), med3i(xe[j] - 1, -2, RC));
                if (I16) { re0[j] = sj[0]; re1[j] = sj[1]; if (GAP == 2) re2[j] = sj[RCS + 1]; }
                else { re0[j] = sj[0]; re1[j] = sj[RCS + 1]; if (GAP == 2) re2[j] = sj[2 * RCS + 1]; }
            }
        }
        if (NPC >= 4) {
            const int pb2 = g2 & 0xfff, pb3 = g3 & 0xfff; Wp2 = (((g2 >> 12) & 0xfff) - pb2 + 1) * PN; Wp3 = (((g3 >> 12) & 0xfff) - pb3 + 1) * PN;
            x2 = colrel - pb2 * PN; x3 = colrel - pb3 * PN;
            const int *s2 = ring_at(__builtin_amdgcn_readlane(vslot, p2), med3i(x2 - 1, -2, RC)), *s3 = ring_at(__builtin_amdgcn_readlane(vslot, p3), med3i(x3 - 1, -2, RC));
            if (I16) { rc0 = s2[0]; rc1 = s2[1]; rd0 = s3[0]; rd1 = s3[1]; if (GAP == 2) { rc2 = s2[RCS + 1]; rd2 = s3[RCS + 1]; } }
            else { rc0 = s2[0]; rc1 = s2[RCS + 1]; rd0 = s3[0]; rd1 = s3[RCS + 1]; if (GAP == 2) { rc2 = s2[2 * RCS + 1]; rd2 = s3[2 * RCS + 1]; } }
        }
        if (NPC >= 2) {
            const int pb1 = g1 & 0xfff; Wp1 = (((g1 >> 12) & 0xfff) - pb1 + 1) * PN;
            x1 = colrel - pb1 * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p1), med3i(x1 - 1, -2, RC));
            if (I16) { rb0 = src[0]; rb1 = src[1]; if (GAP == 2) rb2 = src[RCS + 1]; }
            else { rb0 = src[0]; rb1 = src[RCS + 1]; if (GAP == 2) rb2 = src[2 * RCS + 1]; }
        }
        // work that does not depend on the loaded scores, placed here so that it runs while the LDS reads are in flight (the scheduling
        // barrier keeps the compiler from sinking it behind the wait): band mask, arg-max key constant, ring and arena addresses
        const bool in_band = lane < Wr;
        const int key_c = (vvl == nvr - 1) ? kE : kN;
        const int qd_addr = __builtin_amdgcn_readlane(vslot, ti) + 4 * lane;                                       // LDS byte address of this lane's ring cell
        // arena byte offset of this lane's record -- DIR: of its direction word -- (cur = the row's offset once committed)
        const bool spill = DIR && SLOWV && row_spill;               // (the tight loop's copies never see a row that keeps its records: tile bit 22)
        const unsigned rec_off = DIR ? (unsigned)((cur + (spill ? nvr * CW : 0)) * 32 + lane * DB) : (unsigned)(cur * (int)(PN * sizeof(T)) + lane * (int)(CW * sizeof(T)));
        asm volatile("" :: "v"(key_c), "v"(qd_addr), "v"(rec_off));      // (materialised here, not sunk to their uses)
        __builtin_amdgcn_sched_barrier(0);
        if (I16) { Mv = (int)(short)raw0; E1v = raw1 >> 16; E2v = raw2; } else { Mv = raw0; E1v = raw1; E2v = raw2; }
        const int Mv_first = Mv;                                   // (match flag below: which predecessor supplies the diagonal)
        int kfirst = 1;                                            // 1 + index of the first predecessor that reaches the running maximum of H[.][col-1]
        int kE1 = 1, kE2 = 1;                                      // (DIR) ... of E1 / E2 at this column
        auto merge_pred = [&](int r0_, int r1_, int r2_, int x_, int Wp_, int kidx) __attribute__((always_inline)) {
            int hm1, ev1, ev2 = inf;
            if (I16) { hm1 = (int)(short)r0_; ev1 = r1_ >> 16; ev2 = r2_; } else { hm1 = r0_; ev1 = r1_; ev2 = r2_; }
            const bool inH = (unsigned)x_ < (unsigned)(Wp_ + PN), inE = (unsigned)x_ < (unsigned)Wp_;
            if (NPC >= 4) kfirst = (inH && hm1 > Mv) ? kidx : kfirst;
            if (DIR && NPC >= 4) { kE1 = (inE && ev1 > E1v) ? kidx : kE1; if (GAP == 2) kE2 = (inE && ev2 > E2v) ? kidx : kE2; }      // (two predecessors: read off E afterwards, below)
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        };
        const int E1v_first = E1v, E2v_first = E2v;
        if (NPC >= 2) {
            asm volatile("" : "+v"(rb0), "+v"(rb1));               // (the loads above stay unconditional)
            if (GAP == 2) asm volatile("" : "+v"(rb2));
            merge_pred(rb0, rb1, rb2, x1, Wp1, 2);
        }
        if (NPC >= 4) {
            asm volatile("" : "+v"(rc0), "+v"(rc1), "+v"(rd0), "+v"(rd1));
            if (GAP == 2) asm volatile("" : "+v"(rc2), "+v"(rd2));
            merge_pred(rc0, rc1, rc2, x2, Wp2, 3);
            merge_pred(rd0, rd1, rd2, x3, Wp3, 4);                 // (np == 3: the third predecessor again -- no change, kfirst keeps 3 or less)
        }
        if (NPC == 8) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(re0[j]), "+v"(re1[j])); if (GAP == 2) asm volatile("" : "+v"(re2[j])); }
#pragma unroll
            for (int j = 0; j < 4; ++j) merge_pred(re0[j], re1[j], re2[j], xe[j], We[j], 5 + j);      // (a repeated last predecessor changes nothing: strict > keeps kfirst)
        }
        const int h = Mv + q;                                      // no wrap possible once the check below passes
        int lowest = imin(h, E1v); if (GAP == 2) lowest = imin(lowest, E2v);
        const bool near_wrap = __any(in_band && lowest < fast_lo);      // decided here, acted on after the scan below: the compare runs beside it, the
                                                                        // branch is off the row's dependent chain (nothing is stored before it)
        int hs = h; if (GAP == 2) hs = imax(imax(h, E1v), E2v);
        // lane 0's scan input is first - e (first = H of the band's first column before any E / F merge = h of lane 0): the shift leaves lane 0's
        // own h - e in place, no trip through an SGPR
        // The row arg-max is taken from max(M + q, E) -- an F term is some H of the same row minus at least o + e (reference :870-874 / :990-997), it
        // never holds the row maximum -- so its reduction runs beside the F scan(s), steps interleaved, instead of after H (reference :1043-1057:
        // value, then lowest lane residue, then the end_sn vector, then the lowest vector; columns past the query end never win, :1049-1056)
        const int hsE_ = GAP == 1 ? imax(h, E1v) : hs;
        const bool am_ok = in_band && colrel <= qlen;
        unsigned akey = I16 ? (am_ok ? ((unsigned)hsE_ << 16) + (unsigned)key_c : 0u) : 0u;
        int aval = am_ok ? hsE_ : INT_MIN;
        int s1_ = wave_shr1(h - e1, hs + le1), s2_ = 0;
        if (GAP == 2) s2_ = wave_shr1(h - e2, hs + le2);
        if (I16) { if (GAP == 2) wave_scan3_iiu(s1_, s2_, akey); else wave_scan2_iu(s1_, akey); }
        else { if (GAP == 2) wave_scan3_iii(s1_, s2_, aval); else wave_scan2_ii(s1_, aval); }
        int F1 = imax(s1_ - cf1, inj1), F2 = inf;
        if (GAP == 2) F2 = imax(s2_ - cf2, inj2);
        if (__builtin_expect(near_wrap, 0)) return -1;
        if (SLOWV && end_sn > max_pe) {                             // vectors beyond every predecessor's band: literal masked scan (reference :859-875 / :978-997), as chunk_tail
            const int nfast = max_pe - beg_sn + 1, lastl = nfast * PN - 1;      // 1 <= nfast < nvr
            const int first = __builtin_amdgcn_readlane(imax(s1_, hs + le1), lastl) - lastl * e1;
            int first2 = 0; if (GAP == 2) first2 = __builtin_amdgcn_readlane(imax(s2_, hs + le2), lastl) - lastl * e2;
            T f1t = (T)F1, f2t = (T)F2, fi = (T)first, fi2 = (T)first2;
            slow_f_vectors<T, GAP>(beg_sn, end_sn, max_pe, nfast, l, vvl, (T)hs, (T)inf, (T)e1, (T)oe1, (T)o1, (T)e2, (T)oe2, (T)o2, f1t, f2t, fi, fi2);
            F1 = (int)f1t; F2 = (int)f2t;
        }
        // ---- from here on the row is committed
        const int rec_pn = cur;                                    // (DIR, a row that keeps its records: they start here, the words follow)
        off_pn = cur + (spill ? nvr * CW : 0); cur += row_units(nvr, spill);
        int Hout, E1out, E2out = inf, en1 = 0, t2a = 0, t2b = 0;
        if (GAP == 1) {
            const int tmp = imax(h, E1v);
            Hout = imax(tmp, F1);
            t2a = Hout - oe1; if (DIR) asm("" : "+v"(t2a));       // (kept as a value of its own: the word's uE field is E's maximum minus this term)
            en1 = imax(E1v - e1, t2a);
            E1out = (Hout == tmp) ? en1 : inf;
        } else {
            Hout = imax(hs, imax(F1, F2));
            t2a = Hout - oe1; t2b = Hout - oe2; if (DIR) asm("" : "+v"(t2a), "+v"(t2b));
            E1out = imax(E1v - e1, t2a); E2out = imax(E2v - e2, t2b);
        }
        // record address = arena base + a 32-bit byte offset (an arena is far below 4 GB): one VALU add, no 64-bit pointer arithmetic per row
        T *const H = DIR ? (T *)((char *)io.planes + (size_t)rec_pn * 32) : (T *)((char *)io.planes + (size_t)rec_off) - lane * CW;
        const int he = I16 ? (int)__builtin_amdgcn_perm((unsigned)E1out, (unsigned)Hout, 0x05040100u) : 0;      // H | E1 << 16 (int16: also the score-ring word)
        // match flag for the backtrack (spare slot of the record, finish_alignment PL_FLAG): 1 + index of the first predecessor k (list order) with
        // H[k][col-1] + q == H[col], 0 = none.  Only a predecessor that supplies the maximum Mv can satisfy it, and only when H == Mv + q.
        // (A predecessor value read from outside its band is `inf`: the backtrack re-checks the column range before it trusts the flag.)
        const int mflag = (h == Hout) ? (NPC >= 4 ? kfirst : ((NPC == 2 && Mv != Mv_first) ? 2 : 1)) : 0;
        if constexpr (DIR) {
            // the cell's direction word (dir_plane.h): uE = E's own maximum minus its "opened from H" term = max(o - (H - Ein), 0), dF = min(H - F, cap);
            // where F came from is decided from the left neighbour's word -- except in the vectors of the reference's masked scan, which get the literal test
            const unsigned u1 = (unsigned)((GAP == 1 ? en1 : E1out) - t2a), d1 = umin_((unsigned)(Hout - F1), (unsigned)CAPF1);
            unsigned u2 = 0, d2 = 0;
            if (GAP == 2) { u2 = (unsigned)(E2out - t2b); d2 = umin_((unsigned)(Hout - F2), (unsigned)CAPF2); }
            // kM | kE1 << 4 (| kE2 << 8): one predecessor -- two constants
            if (NPC == 2) { kE1 = E1v != E1v_first ? 2 : 1; if (GAP == 2) kE2 = E2v != E2v_first ? 2 : 1; }      // (the second predecessor holds the maximum iff it raised it: strictly greater)
            const unsigned kf = NPC == 1 ? ((h == Hout) ? (GAP == 1 ? 0x11u : 0x111u) : (GAP == 1 ? 0x10u : 0x110u))
                                         : (GAP == 1 ? ((unsigned)kE1 << 4) | (unsigned)mflag : ((((unsigned)kE2 << 4) | (unsigned)kE1) << 4) | (unsigned)mflag);
            unsigned wd = dir_word<GAP>(kf, u1, u2, d1, d2);
            if (SLOWV && end_sn > max_pe) {
                const int nfast_ = max_pe - beg_sn + 1;
                const int Hm1 = wave_shr1(Hout, Hout), F1m1 = wave_shr1(F1, F1);
                unsigned l1 = dir_literal<T>(Hm1, F1m1, F1, oe1, e1), l2 = 0;
                if (GAP == 2) { const int F2m1 = wave_shr1(F2, F2); l2 = dir_literal<T>(Hm1, F2m1, F2, oe2, e2); }
                // (lane 0 of the band -- no stored left neighbour -- is never a masked-scan vector here: nfast_ >= 1)
                if (vvl >= nfast_) wd |= GAP == 1 ? l1 << DIRA_LF1_SH : (l1 << DIRC_LF1_SH | l2 << DIRC_LF2_SH);
            }
            char *const dp = (char *)io.planes + (size_t)rec_off;
            if (GAP == 1) *(uint16_t *)dp = (uint16_t)wd; else *(uint32_t *)dp = wd;
        }
        if (DIR && !spill) {}
        else if (DIR && !in_band) {}      // (the row's words start right behind its last record)
        else if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(H + lane * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(H + lane * CW); dst[0] = r0; dst[1] = r1; }
        {
            int *qd = (int *)ring_at(qd_addr, 0);
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
            else { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS] = in_band ? E2out : inf; }
            qd[64] = infw; if (NPW > 1) qd[RCS + 64] = inf; if (NPW > 2) qd[2 * RCS + 64] = inf;      // (RC <= 128 for these rows: tv_meta bit 17)
        }
        // ---- arg-max (keys reduced above)
        if (I16) {
            const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)akey, 63);
            rowmax = (int)(kb >> 16) - 32768;
            mi = (rowmax > inf) ? beg_sn * PN + (int)(kb & 63) : -1;      // the winning lane IS the column offset
        } else {
            const int vmax = __builtin_amdgcn_readlane(aval, 63);
            rowmax = vmax;
            const unsigned key = (am_ok && hsE_ == vmax) ? (((unsigned)(PN - 1 - l) << 12) | (unsigned)((vvl == nvr - 1) ? 8 : NV - 1 - vvl)) : 0u;
            const unsigned kb = wave_max_u32_s(key);
            const int vrel = (kb & 8) ? nvr - 1 : NV - 1 - (int)(kb & 7);
            mi = (vmax > inf) ? (beg_sn + vrel) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)) : -1;
        }
        return 1;
    };
    auto turbo_body = [&](auto npc, auto slowc, int row, int ti) __attribute__((always_inline)) -> int {
        if constexpr (GAP == 0) return turbo_lin(npc, slowc, row, ti); else return turbo_aff(npc, slowc, row, ti);
    };

    constexpr int NWP = NW <= 2 ? 2 : (NW <= 4 ? 4 : 8);            // exchange entries read per lane group
    // ---- MULTI-CHUNK body for wide bands (10 kb reads: 240-300 columns = 4-5 chunks of 64; where a read drifts against the graph the band
    //      opens to 6, seldom 7 chunks for thousands of rows -- three instantiations: 2-3, 4-5 and 6-7 chunks): every chunk of the row is in registers at once.
    //      The lane owns column (64 c + lane) of each chunk c, so the chunks are independent instruction streams that the scheduler interleaves
    //      (LDS reads of all chunks in flight together, DPP scans of all chunks back to back without wait states), and everything that is
    //      per row -- band, conditions, arg-max reduction, commit -- is paid once for ~250 columns instead of once per 64.
    //      F: per-chunk UNSEEDED prefix maxima of g = hs + lane * e, then a scalar carry chain over the chunk totals:
    //      seed[0] = first - e, seed[c+1] = max(total[c], seed[c]) - 64 e, F = max(S, seed) - cf  (chunk_tail's closed form, carried in H units).
    //      The arg-max is taken from max(M + q, E): an F term is some H of the same row minus at least o + e (reference :870-874 / :990-997).
    //      ilp_band: 0 = not applicable, -2 = arena overflow, else the number of chunks; ilp_chunks: 0 = not applicable (nothing touched), 1 = done.
    //      TEAMS (NW > 1 wavefronts per alignment): every wavefront runs the same row loop on its own copy of the per-row registers and takes a
    //      contiguous share of the row's chunks (c0 .. c0 + cnt - 1); ONE exchange through LDS per row carries each wavefront's carry-chain result
    //      (seed out of its last chunk, computed as if nothing came in: the chain is max-plus, the incoming seed is folded in afterwards), its
    //      arg-max key and its wrap flag; a second barrier at the end of the row publishes the ring slot.
    constexpr int NCHX = XL ? 11 : 7;      // (XL: the long-read form of the wide kernel, 704-column ring, one wavefront per SIMD's worth of registers)
    constexpr bool TEAM = NW > 1;
    int two_chunk_streak = 0;                                       // (narrow kernel) the last row took the two-chunk body with a band of 65-128 columns: the next one tries it first
    int qcx_beg_sn = -1, qcx_c0 = -1, qoffx[NCHX] = {};            // cached query codes of this lane's column in every chunk of this wavefront, for band start qcx_beg_sn
    int ilp_far = 0;                                                // bit k: predecessor k of the row is not in the score ring (older than its depth, or a row too wide for it): HBM gather
    auto ilp_band = [&](int row, int ti) __attribute__((always_inline)) -> int {
        int mn_mi, mx_mi, min_pb;
        ilp_far = 0;
        auto is_far = [&](int p, int g_) __attribute__((always_inline)) { return (row - p >= RR || !(g_ & GEO_RING)) ? 1 : 0; };
        {
            const int p = __builtin_amdgcn_readlane(tv_p0, ti), g_ = __builtin_amdgcn_readlane(vg_geo, p & 63);
            mn_mi = mx_mi = __builtin_amdgcn_readlane(vg_mi, p & 63); min_pb = g_ & 0xfff; max_pe = (g_ >> 12) & 0xfff; ilp_far = is_far(p, g_);
        }
        auto more = [&](int tvp, int k) __attribute__((always_inline)) {
            const int p = __builtin_amdgcn_readlane(tvp, ti), g_ = __builtin_amdgcn_readlane(vg_geo, p & 63), mi_ = __builtin_amdgcn_readlane(vg_mi, p & 63);
            mn_mi = imin(mn_mi, mi_); mx_mi = imax(mx_mi, mi_); min_pb = imin(min_pb, g_ & 0xfff); max_pe = imax(max_pe, (g_ >> 12) & 0xfff); ilp_far |= is_far(p, g_) << k;
        };
        if (np > 1) { more(tv_p1, 1); if (np > 2) { more(tv_p2, 2); if (np > 3) { more(tv_p3, 3); if (np > 4) { more(tv_p4, 4); if (np > 5) { more(tv_p5, 5); if (np > 6) { more(tv_p6, 6);
                if (np > 7) more(tv_p7, 7); } } } } } }
        set_band(std::true_type{}, mn_mi, mx_mi, min_pb);
        const int nvr = end_sn - beg_sn + 1, Wr = nvr * PN, nch = (Wr + 63) >> 6;
        bool ok = nch <= NCHX && Wr <= RC && max_pe >= beg_sn;
        if (!ok) { WCOUNT(nch > NCHX ? 3 : 2); return 0; }
        // vectors beyond every predecessor's band (literal masked scan) must all sit in the row's last chunk
        if (end_sn > max_pe) ok = ok && ((max_pe + 1 - beg_sn) / NV == nch - 1);
        if (!ok) { WCOUNT(4); return 0; }
        if (cur + row_units(nvr, row_spill) > cap_pn) return -2;
        return nch;
    };
    auto ilp_chunks = [&](auto nchc, int nch, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr int NCH = decltype(nchc)::value;
        const int Wr = (end_sn - beg_sn + 1) * PN;
        // this wavefront's chunks: c0 .. c0 + cnt - 1 (one wavefront: all of them; teams: nch / NW each, the first nch % NW one more)
        int c0 = 0, cnt = nch;
        if (TEAM) { const int bs_ = nch / NW, rm_ = nch - bs_ * NW; cnt = bs_ + (wid < rm_ ? 1 : 0); c0 = wid * bs_ + imin(wid, rm_); }
        const int colb = beg_sn * PN + lane + 64 * c0;              // this lane's column in the wavefront's first chunk
        if (__builtin_expect(beg_sn != qcx_beg_sn || (TEAM && c0 != qcx_c0), 0)) {
            qcx_beg_sn = beg_sn; qcx_c0 = c0;
#pragma unroll
            for (int c = 0; c < NCHX; ++c) { const int col = colb + 64 * c; qoffx[c] = (col >= 1 && col <= qlen) ? qat(col - 1) : m; }
        }
        FSTAMP(0)
        const int *mrow = s_mx + base * m1;
        int q[NCH], Mv[NCH], E1v[NCH], E2v[NCH], kb[NCH];
        int kE1[NCH], kE2[NCH];                                     // (DIR) 1 + list index of the first predecessor holding the maximum E1 / E2 of the column
#pragma unroll
        for (int c = 0; c < NCH; ++c) q[c] = mrow[qoffx[c]];
        // ---- predecessor gather: the first one unmasked (guard cells and padding of the ring row yield what the reference reads), the others
        //      with one unsigned range compare per plane; the first two predecessors' reads go out together
        auto ring_read = [&](int p, int g_, int c, int &hm1, int &ev1, int &ev2) __attribute__((always_inline)) {
            const int x = colb + 64 * c - (g_ & 0xfff) * PN;
            const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), med3i(x - 1, -2, RC));
            ev2 = inf;
            if (I16) { const int w0 = src[0], w1 = src[1]; hm1 = (int)(short)w0; ev1 = w1 >> 16; if (GAP == 2) ev2 = src[RCS + 1]; }
            else if (EPACK) { hm1 = src[0]; const int h0 = src[1]; const unsigned dd = (unsigned)src[RCS + 1]; ev1 = h0 - (int)(dd & 0xffffu); ev2 = h0 - (int)(dd >> 16); }
            else { hm1 = src[0]; ev1 = src[RCS + 1]; if (GAP == 2) ev2 = src[2 * RCS + 1]; }
        };
        auto merge = [&](int g_, int c, int hm1, int ev1, int ev2, int kidx) __attribute__((always_inline)) {
            const int pb = g_ & 0xfff, Wp = (((g_ >> 12) & 0xfff) - pb + 1) * PN, x = colb + 64 * c - pb * PN;
            const bool inH = (unsigned)x < (unsigned)(Wp + PN), inE = (unsigned)x < (unsigned)Wp;
            kb[c] = (inH && hm1 > Mv[c]) ? kidx : kb[c];
            if (DIR) { kE1[c] = (inE && ev1 > E1v[c]) ? kidx : kE1[c]; if (GAP == 2) kE2[c] = (inE && ev2 > E2v[c]) ? kidx : kE2[c]; }
            Mv[c] = inH ? imax(Mv[c], hm1) : Mv[c]; E1v[c] = inE ? imax(E1v[c], ev1) : E1v[c]; if (GAP == 2) E2v[c] = inE ? imax(E2v[c], ev2) : E2v[c];
        };
        // a predecessor outside the ring: its cells come from the arena in HBM (records this wave stored at least RR rows ago), every
        // chunk's loads in flight together; outside its band the reference reads / assigns "inf", exactly what the ring's guards and padding give
        auto hbm_read_all = [&](int p, int g_, int *hc, int *ec1, int *ec2) __attribute__((always_inline)) {
            const int pb = g_ & 0xfff, Wp = (((g_ >> 12) & 0xfff) - pb + 1) * PN;
            // (DIR: the predecessor kept its score records -- tile bit 21 / a row too wide for the ring -- in front of its direction words)
            const T *Hp = io.planes + (long long)(uint32_t)(__builtin_amdgcn_readlane(vg_off, p & 63) - (DIR ? (((g_ >> 12) & 0xfff) - pb + 1) * CWR : 0)) * PN;
            gld_wait();                                              // (earlier score-plane stores of this wave are complete)
            if (TEAM) lds_barrier();                                 // (... and of the other wavefronts of the team: the far flag is the same in all of them)
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int x = colb + 64 * c - pb * PN, xh = med3i(x - 1, 0, Wp - 1), xe = med3i(x, 0, Wp - 1);
                // H[x-1]; H[x], differences
                if constexpr (CPK) { gld_async_cell(hc[c], Hp + (long long)xh * CWR); gld_async_cell(ec1[c], Hp + (long long)xe * CWR); gld_async_cell(ec2[c], Hp + (long long)xe * CWR + 1); }
                else {
                    gld_async_cell(hc[c], Hp + (long long)xh * CWR); gld_async_cell(ec1[c], Hp + (long long)xe * CWR + PL_E1);
                    if (GAP == 2) gld_async_cell(ec2[c], Hp + (long long)xe * CWR + PL_E2); else ec2[c] = inf;
                }
            }
            // (the wait names the loaded registers: a register-only use must not be scheduled above it)
#pragma unroll
            for (int c = 0; c < NCH; ++c) { if (GAP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]), "+v"(ec2[c]) :: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]) :: "memory"); }
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int x = colb + 64 * c - pb * PN;
                if constexpr (CPK) { const int h0 = ec1[c]; const unsigned d1_ = (unsigned)ec2[c] & 0xffffu, d2_ = (unsigned)ec2[c] >> 16; ec1[c] = d1_ == 0xffffu ? inf : h0 - (int)d1_;
                        ec2[c] = d2_ == 0xffffu ? inf : h0 - (int)d2_; }
                hc[c] = (unsigned)(x - 1) < (unsigned)Wp ? hc[c] : inf; ec1[c] = (unsigned)x < (unsigned)Wp ? ec1[c] : inf; if (GAP == 2) ec2[c] = (unsigned)x < (unsigned)Wp ? ec2[c] : inf;
            }
        };
        auto gather_pred = [&](int k, int tvp) __attribute__((always_inline)) {
            const int p = __builtin_amdgcn_readlane(tvp, ti), g_ = __builtin_amdgcn_readlane(vg_geo, p & 63);
            int hc[NCH], ec1[NCH], ec2[NCH];
            if ((ilp_far >> k) & 1) hbm_read_all(p, g_, hc, ec1, ec2);
            else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) ring_read(p, g_, c, hc[c], ec1[c], ec2[c]);
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c) { if (k == 0) { Mv[c] = hc[c]; E1v[c] = ec1[c]; E2v[c] = ec2[c]; kb[c] = 1; kE1[c] = 1; kE2[c] = 1; } else merge(g_, c, hc[c], ec1[c], ec2[c], k + 1); }
        };
        {
            const int p0 = __builtin_amdgcn_readlane(tv_p0, ti), g0 = __builtin_amdgcn_readlane(vg_geo, p0 & 63);
            if (np == 1 && !ilp_far) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) { ring_read(p0, g0, c, Mv[c], E1v[c], E2v[c]); kb[c] = 1; kE1[c] = 1; kE2[c] = 1; }
            } else if (np == 1) gather_pred(0, tv_p0);
            else {
                if (!(ilp_far & 3)) {                               // the first two predecessors' ring reads go out together
                    const int p1 = __builtin_amdgcn_readlane(tv_p1, ti), g1_ = __builtin_amdgcn_readlane(vg_geo, p1 & 63);
                    int hb[NCH], eb1[NCH], eb2[NCH];
#pragma unroll
                    for (int c = 0; c < NCH; ++c) { ring_read(p0, g0, c, Mv[c], E1v[c], E2v[c]); kb[c] = 1; kE1[c] = 1; kE2[c] = 1; ring_read(p1, g1_, c, hb[c], eb1[c], eb2[c]); }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) merge(g1_, c, hb[c], eb1[c], eb2[c], 2);
                } else { gather_pred(0, tv_p0); gather_pred(1, tv_p1); }
                // predecessors 3..8: ONE copy of the gather code in a run-time loop (code size: the kernel has to live in the 64 KB instruction cache that
                // two CUs share, and on real graphs the wavefronts of a CU pair are at different places of it)
                for (int k = 2; k < np; ++k) {
                    const int tvk = k == 2 ? tv_p2 : (k == 3 ? tv_p3 : (k == 4 ? tv_p4 : (k == 5 ? tv_p5 : (k == 6 ? tv_p6 : tv_p7))));
                    gather_pred(k, tvk);
                }
            }
        }
        FSTAMP(1)
        // ---- H before F, wrap guard of the closed form (lanes of vectors <= max_pre_end_sn, as chunk_tail)
        //      (checked over ALL lanes: lanes outside the closed-form vectors hold inf + q, which is above the limit unless the penalties are
        //       extreme -- then the row simply takes the exact bodies)
        int h[NCH], hs[NCH], hsE[NCH], lowest = INT_MAX;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            h[c] = wr(Mv[c] + q[c]);
            hs[c] = h[c]; if (GAP == 2) hs[c] = imax(imax(h[c], E1v[c]), E2v[c]);
            hsE[c] = GAP == 1 ? imax(h[c], E1v[c]) : hs[c];
            lowest = imin(lowest, h[c]);
        }
        const bool wrap_here = __any(lowest < fast_lo);
        if (!TEAM && __builtin_expect(wrap_here, 0)) return 0;
        // ---- unseeded prefix maxima per chunk, all chains interleaved
        int g1[NCH], g2[NCH], s1[NCH], s2[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) { g1[c] = hs[c] + le1; s1[c] = wave_shr1(INT_MIN, g1[c]); if (GAP == 2) { g2[c] = hs[c] + le2; s2[c] = wave_shr1(INT_MIN, g2[c]); } }
        // ---- arg-max candidates (reference :1043-1057): one packed key per lane, one reduction.  int16: chunk_tail's key.  int32: the value is
        //      taken relative to floor = (maximum of the first predecessor's row) - 2^19, 21 bits, above residue / end-vector / vector-order bits;
        //      a winner outside (0, 2^21 - 1) -- never seen on real scores, consecutive rows differ by a few units -- sends the row to the exact bodies.
        unsigned amk = 0;
        const int vfloor = I16 ? 0 : __builtin_amdgcn_readlane(vg_vm, __builtin_amdgcn_readlane(tv_p0, ti) & 63) - (1 << 19);
        const int relv_end = end_sn - beg_sn;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int cg = c0 + c, vb = beg_sn + cg * NV;          // chunk index in the row
            // (the end vector is in the last chunk)
            const bool in_band = (!TEAM && c < NCH - 2) ? true : cg * 64 + lane < Wr, is_end = (!TEAM && c < NCH - 2) ? false : (cg * NV + vvl == relv_end);
            int cand = hsE[c]; if (end_sn == qlen_sn) cand = (is_end && colb + 64 * c > qlen) ? inf : cand;
            unsigned key;
            if (I16) key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
            else key = ((unsigned)imin(imax(cand, vfloor) - vfloor, 0x1FFFFF) << 11) | (unsigned)(ktie - cg * NV) | (is_end ? 64u : 0u);      // (max first: inf - floor must not wrap)
            amk = (in_band && key > amk) ? key : amk;
        }
        // interleaved DPP chains: F scans of every chunk + the arg-max key
        {
            auto step = [&](auto ctrl, auto rmask) __attribute__((always_inline)) {
                constexpr int CT = decltype(ctrl)::value, RM = decltype(rmask)::value;
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    s1[c] = imax(s1[c], __builtin_amdgcn_update_dpp(INT_MIN, s1[c], CT, RM, 0xF, false));
                    if (GAP == 2) s2[c] = imax(s2[c], __builtin_amdgcn_update_dpp(INT_MIN, s2[c], CT, RM, 0xF, false));
                }
                const unsigned t = (unsigned)__builtin_amdgcn_update_dpp(0, (int)amk, CT, RM, 0xF, false); amk = t > amk ? t : amk;
            };
            step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xF>{});
            step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xF>{});
            step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xF>{});
            step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xF>{});
            step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});
            step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});
        }
        unsigned kbst = (unsigned)__builtin_amdgcn_readlane((int)amk, 63);
        FSTAMP(2)
        // ---- carry chain over the chunk totals (scalar), then F of every chunk.  seed[c] = "first - e" of chunk c; seed[c + 1] = max(total[c], seed[c]) - 64 e.
        //      Wavefront 0 (or the only one) starts from the row's first column; the others start from "nothing" (INT_MIN: total[c] always wins the
        //      max, so nothing wraps) and fold the seed that comes in from the wavefronts before them after the exchange.
        //      (one wavefront: the chain runs in the vector unit on wave-uniform values -- the totals are read out of lane 63 back to back and the
        //       dependent max / subtract steps need no trip through the scalar unit)
        int seed1[NCH + 1], seed2[NCH + 1];
        seed1[0] = (!TEAM || wid == 0) ? __builtin_amdgcn_readlane(h[0], 0) - e1 : INT_MIN; seed2[0] = (!TEAM || wid == 0) ? seed1[0] + e1 - e2 : INT_MIN;
        if (!TEAM) { asm("" : "+v"(seed1[0])); if (GAP == 2) asm("" : "+v"(seed2[0])); }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            seed1[c + 1] = imax(__builtin_amdgcn_readlane(imax(s1[c], g1[c]), 63), seed1[c]) - 64 * e1;
            if (GAP == 2) seed2[c + 1] = imax(__builtin_amdgcn_readlane(imax(s2[c], g2[c]), 63), seed2[c]) - 64 * e2; else seed2[c + 1] = INT_MIN;
        }
        if constexpr (TEAM) {
            // exchange entry of this wavefront: {chain result out of its last chunk (two planes), arg-max key, wrap flag}
            int out1 = seed1[0], out2 = seed2[0];
#pragma unroll
            for (int c = 0; c < NCH; ++c) if (c < cnt) { out1 = seed1[c + 1]; out2 = seed2[c + 1]; }
            int4 *xs = xch + (row & 1) * 8;
            if (lane == 0) xs[wid] = make_int4(out1, out2, cnt > 0 ? (int)kbst : 0, wrap_here && cnt > 0 ? 1 : 0);
            lds_barrier();
            const int4 en = xs[lane & (NWP - 1)];
            // incoming seed: fold the entries of the wavefronts before this one in order (saturating: INT_MIN stays "nothing")
            int in1 = INT_MIN, in2 = INT_MIN, anywrap = 0; unsigned kall = 0;
#pragma unroll
            for (int j = 0; j < NW; ++j) {
                const int a1 = __builtin_amdgcn_readlane(en.x, j), a2 = __builtin_amdgcn_readlane(en.y, j);
                const unsigned kj = (unsigned)__builtin_amdgcn_readlane(en.z, j);
                anywrap |= __builtin_amdgcn_readlane(en.w, j); kall = kj > kall ? kj : kall;
                if (j < wid) {
                    const int bs_ = nch / NW, rm_ = nch - bs_ * NW, cj = (bs_ + (j < rm_ ? 1 : 0)) * 64;
                    in1 = imax(a1, imax(in1, INT_MIN + cj * e1) - cj * e1); in2 = imax(a2, imax(in2, INT_MIN + cj * e2) - cj * e2);
                }
            }
            if (__builtin_expect(anywrap, 0)) return 0;
            kbst = kall;
            if (wid > 0) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    seed1[c] = imax(seed1[c], imax(in1, INT_MIN + c * 64 * e1) - c * 64 * e1);
                    if (GAP == 2) seed2[c] = imax(seed2[c], imax(in2, INT_MIN + c * 64 * e2) - c * 64 * e2);
                }
            }
        }
        if (!I16) { const unsigned tv = kbst >> 11; if (__builtin_expect(tv == 0u || tv == 0x1FFFFFu, 0)) return 0; }
        FSTAMP(3)
        // ---- from here on the row is committed
        T *const Hrow = io.planes + (long long)cur * PN + (long long)(lane + 64 * c0) * CWR;
        off_pn = cur + ((DIR && row_spill) ? (end_sn - beg_sn + 1) * CWR : 0); cur += row_units(end_sn - beg_sn + 1, row_spill);
        char *const Drow = (char *)io.planes + (size_t)off_pn * 32 + (size_t)(lane + 64 * c0) * DB;      // (DIR) this lane's direction word in the wavefront's first chunk
        int *const qd = (int *)ring_at(__builtin_amdgcn_readlane(vslot, ti) + 4 * (lane + 64 * c0), 0);
        // (nch is NCH - 1 or NCH: chunks 0 .. NCH - 3 are full, only the last two need band masks, only the last one a store guard)
        int F1[NCH], F2[NCH], S1[NCH], S2[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            S1[c] = imax(s1[c], seed1[c]); F1[c] = imax(S1[c] - cf1, inj1); F2[c] = inf; S2[c] = 0;
            if (GAP == 2) { S2[c] = imax(s2[c], seed2[c]); F2[c] = imax(S2[c] - cf2, inj2); }
        }
        if (__builtin_expect(end_sn > max_pe, 0)) {                  // vectors beyond every predecessor's band: literal masked scan, last chunk only
#pragma unroll
            for (int c = TEAM ? 0 : NCH - 2; c < NCH; ++c) if (c0 + c == nch - 1) {
                const int vb = beg_sn + (c0 + c) * NV;
                const int nvec = imin(NV, end_sn - vb + 1), nfast = imax(0, imin(nvec, max_pe - vb + 1));
                int first, first2 = 0;
                if (nfast > 0) {
                    const int lastl = nfast * PN - 1;
                    first = __builtin_amdgcn_readlane(imax(S1[c], g1[c]), lastl) - lastl * e1;
                    if (GAP == 2) first2 = __builtin_amdgcn_readlane(imax(S2[c], g2[c]), lastl) - lastl * e2;
                } else { first = __builtin_amdgcn_readfirstlane(seed1[c]) + e1; if (GAP == 2) first2 = __builtin_amdgcn_readfirstlane(seed2[c]) + e2; }
                T f1t = (T)F1[c], f2t = (T)F2[c], fi = (T)first, fi2 = (T)first2;
                slow_f_vectors<T, GAP>(vb, end_sn, max_pe, nfast, l, vvl, (T)hs[c], (T)inf, (T)e1, (T)oe1, (T)o1, (T)e2, (T)oe2, (T)o2, f1t, f2t, fi, fi2);
                F1[c] = (int)f1t; F2[c] = (int)f2t;
            }
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const bool in_band = (!TEAM && c < NCH - 2) ? true : (c0 + c) * 64 + lane < Wr;
            int Hout, E1out, E2out = inf, t2a, t2b = 0, en_a = 0;
            if (GAP == 1) {
                Hout = imax(hsE[c], F1[c]);
                t2a = wr(Hout - oe1); if (DIR) asm("" : "+v"(t2a));      // (kept as a value of its own: the word's uE field is E's maximum minus this term)
                en_a = imax(wr(E1v[c] - e1), t2a);
                E1out = (Hout == hsE[c]) ? en_a : inf;
            } else {
                Hout = imax(hs[c], imax(F1[c], F2[c]));
                t2a = wr(Hout - oe1); t2b = wr(Hout - oe2); if (DIR) asm("" : "+v"(t2a), "+v"(t2b));
                E1out = imax(wr(E1v[c] - e1), t2a);
                E2out = imax(wr(E2v[c] - e2), t2b);
            }
            const int mflag = (Mv[c] + q[c] == Hout && kb[c] <= 64) ? kb[c] : 0;
            const int he = (int)(((unsigned)Hout & 0xffffu) | ((unsigned)E1out << 16));
            if constexpr (DIR) if (c < NCH - 1 || nch == NCH) {
                // direction word of the column (dir_plane.h; as in the straight-line body of the narrow loop)
                const unsigned u1 = (unsigned)((GAP == 1 ? en_a : E1out) - t2a), d1 = umin_((unsigned)(Hout - F1[c]), (unsigned)CAPF1);
                unsigned u2 = 0, d2 = 0;
                if (GAP == 2) { u2 = (unsigned)(E2out - t2b); d2 = umin_((unsigned)(Hout - F2[c]), (unsigned)CAPF2); }
                unsigned wd = dir_pack<GAP>((unsigned)mflag, (unsigned)kE1[c], (unsigned)kE2[c], u1, u2, d1, d2);
                if (c >= NCH - 2) if (__builtin_expect(end_sn > max_pe, 0) && c == nch - 1) {      // the vectors of the reference's masked F scan (last chunk only): the literal test
                    const int vb = beg_sn + c * NV, nfast_ = imax(0, imin(imin(NV, end_sn - vb + 1), max_pe - vb + 1));
                    int pH = Hout, pF1 = F1[c], pF2 = F2[c];                             // left neighbour of the chunk's first column: the previous chunk's last one
                    if (c > 0) {
                        const int Hp_ = GAP == 1 ? imax(hsE[c > 0 ? c - 1 : 0], F1[c > 0 ? c - 1 : 0]) : imax(hs[c > 0 ? c - 1 : 0], imax(F1[c > 0 ? c - 1 : 0], F2[c > 0 ? c - 1 : 0]));
                        pH = __builtin_amdgcn_readlane(Hp_, 63); pF1 = __builtin_amdgcn_readlane(F1[c > 0 ? c - 1 : 0], 63); pF2 = __builtin_amdgcn_readlane(F2[c > 0 ? c - 1 : 0], 63);
                    }
                    const int Hm1 = wave_shr1(pH, Hout), F1m1 = wave_shr1(pF1, F1[c]);
                    unsigned l1 = dir_literal<T>(Hm1, F1m1, F1[c], oe1, e1), l2 = 0;
                    if (GAP == 2) { const int F2m1 = wave_shr1(pF2, F2[c]); l2 = dir_literal<T>(Hm1, F2m1, F2[c], oe2, e2); }
                    if (vvl >= nfast_ && (c > 0 || lane > 0)) wd |= GAP == 1 ? l1 << DIRA_LF1_SH : (l1 << DIRC_LF1_SH | l2 << DIRC_LF2_SH);
                }
                char *const dp = Drow + c * 64 * DB;
                if (GAP == 1) *(uint16_t *)dp = (uint16_t)wd; else *(uint32_t *)dp = wd;
            }
            if (DIR && !row_spill) {}
            else if (DIR && !in_band) {}      // (the row's words start right behind its last record)
            // (teams: in-band lanes only -- another wavefront owns the cells behind the row's end) one record store per lane, all 64 lanes (lanes past the band write cells the next row overwrites:
            //  same wave, program order)
            else if (TEAM ? (c < cnt && in_band) : (c < NCH - 1 || nch == NCH)) {
                T *H = Hrow + c * 64 * CWR;
                if (CSP) {      // compact records (CWR)
                    if (I16 && GAP == 1) *(int *)H = he;
                    else if (I16) { int2 rec; rec.x = he; rec.y = E2out & 0xffff; *(int2 *)H = rec; }
                    else if (GAP == 1) { int2 rec; rec.x = Hout; rec.y = E1out; *(int2 *)H = rec; }
                    else { int2 rec; rec.x = Hout; rec.y = (int)((unsigned)(Hout - E1out) | ((unsigned)(Hout - E2out) << 16)); *(int2 *)H = rec; }
                }
                else if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1[c], 0x05040100u); *(int2 *)H = rec; }
                else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1[c] << 16)); rec.z = F2[c] & 0xffff; rec.w = mflag; *(int4 *)H = rec; }
                else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1[c]; rec.w = mflag; *(int4 *)H = rec; }
                else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1[c]; r1.x = F2[c]; r1.y = mflag; r1.z = 0; r1.w = 0; ((int4 *)H)[0] = r0; ((int4 *)H)[1] = r1; }
            }
            if (!TEAM || c < cnt) {
                if (I16) { qd[c * 64] = in_band ? he : infw; if (GAP == 2) qd[RCS + c * 64] = in_band ? E2out : inf; }
                else if (EPACK) { qd[c * 64] = in_band ? Hout : inf; qd[RCS + c * 64] = in_band ? (int)((unsigned)(Hout - E1out) | ((unsigned)(Hout - E2out) << 16)) : 0; }
                else { qd[c * 64] = in_band ? Hout : inf; qd[RCS + c * 64] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS + c * 64] = in_band ? E2out : inf; }
            }
        }
        // "inf" up to the ring width
        if (!TEAM) { for (int c = NCH; c < (RC >> 6); ++c) { qd[c * 64] = infw; if (NPW > 1) qd[RCS + c * 64] = EPACK ? 0 : inf; if (NPW > 2) qd[2 * RCS + c * 64] = inf; } }
        else {                                                       // (teams: chunk c of the padding is written by wavefront c % NW)
            int *const qrow = (int *)ring_at(__builtin_amdgcn_readlane(vslot, ti) + 4 * lane, 0);
            for (int c = nch; c < (RC >> 6); ++c) if (c % NW == wid) { qrow[c * 64] = infw; if (NPW > 1) qrow[RCS + c * 64] = EPACK ? 0 : inf; if (NPW > 2) qrow[2 * RCS + c * 64] = inf; }
        }
        FSTAMP(4)
        // ---- row arg-max
        mi = -1;
        if (I16) {
            rowmax = (int)(kbst >> 16) - 32768;
            if (rowmax > inf) { mi = (2047 - (int)(kbst & 0x7ff)) * PN + (PN - 1 - (int)((kbst >> 12) & 0xf)); if (mi > qlen) mi = -1; }
        } else {
            rowmax = vfloor + (int)(kbst >> 11);
            if (rowmax > inf) { mi = (beg_sn + 63 - (int)(kbst & 63)) * PN + (PN - 1 - (int)((kbst >> 7) & 0xf)); if (mi > qlen) mi = -1; }
        }
        return 1;
    };

    // ---- FAST body: NP (1, 2, or up to 4 with run-time count) predecessors, all in the 64-row geometry ring and the score ring.
    //      Returns 0 = not applicable (nothing touched), 1 = done, 2 = arena overflow.
    auto fast_body = [&](auto npc, int row, int ti) __attribute__((always_inline)) -> int {
        constexpr int NPC = decltype(npc)::value;
        int pr[4], pgeo[4];
        pr[0] = __builtin_amdgcn_readlane(tv_p0, ti);
        pgeo[0] = __builtin_amdgcn_readlane(vg_geo, pr[0] & 63);
        int mn_mi = __builtin_amdgcn_readlane(vg_mi, pr[0] & 63), mx_mi = mn_mi, min_pb = pgeo[0] & 0xfff, allring = pgeo[0];
        max_pe = (pgeo[0] >> 12) & 0xfff;
        auto more = [&](int k, int tvp) __attribute__((always_inline)) {
            pr[k] = __builtin_amdgcn_readlane(tvp, ti); pgeo[k] = __builtin_amdgcn_readlane(vg_geo, pr[k] & 63);
            const int mi_ = __builtin_amdgcn_readlane(vg_mi, pr[k] & 63);
            mn_mi = imin(mn_mi, mi_); mx_mi = imax(mx_mi, mi_); min_pb = imin(min_pb, pgeo[k] & 0xfff); max_pe = imax(max_pe, (pgeo[k] >> 12) & 0xfff); allring &= pgeo[k];
        };
        if (NPC >= 2) more(1, tv_p1);
        if (NPC >= 4) { pr[2] = pr[1]; pgeo[2] = pgeo[1]; pr[3] = pr[1]; pgeo[3] = pgeo[1]; if (np > 2) more(2, tv_p2); if (np > 3) more(3, tv_p3); }
        set_band(std::true_type{}, mn_mi, mx_mi, min_pb);
        const int Wr = (end_sn - beg_sn + 1) * PN;
        if (!(allring & GEO_RING) || Wr > RC) return 0;
        FSTAMP(0)
        if (!reserve()) return 2;
        to_ring = true;
        T *H = io.planes + (long long)off_pn * PN;
        const int my_slot = (row & (RR - 1)) * (NPW * RCS);
        const int nch = (Wr + 63) >> 6;
        refresh_qc();
        const int *mrow = s_mx + base * m1;
        int first = 0, first2 = 0;
        for (int c = 0; c < nch; ++c) {
            const int col = beg_sn * PN + c * 64 + lane;
            int qc = c == 0 ? qoff0 : qoff1;
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? qat(col - 1) : m;
            const int q = mrow[qc];
            int Mv = lane, E1v = inf, E2v = inf, kb = 0, kE1 = 1, kE2 = 1;
            if (!ABL(16)) from_ring(0, pr[0], pgeo[0], col, Mv, E1v, E2v, kb, 1, kE1, kE2, q);
            if (NPC >= 2 && !ABL(16)) from_ring(1, pr[1], pgeo[1], col, Mv, E1v, E2v, kb, 2, kE1, kE2, q);
            if (NPC >= 4) { if (np > 2) from_ring(2, pr[2], pgeo[2], col, Mv, E1v, E2v, kb, 3, kE1, kE2, q); if (np > 3) from_ring(3, pr[3], pgeo[3], col, Mv, E1v, E2v, kb, 4, kE1, kE2, q); }
            FSTAMP(1)
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot, kE1, kE2);
        }
        pad_ring(nch, my_slot);
        return 1;
    };

    // ---- GENERAL body: any number of predecessors, any distance (HBM copies of geometry and score rows), any band width.
    //      Returns 1 = done, 2 = arena overflow.
    auto general_body = [&](int row, int ti) __attribute__((always_inline)) -> int {
        const int ps = __builtin_amdgcn_readlane(tv_ps, ti);
        int mn_mi = gn, mx_mi = -1, min_pb = 4095; max_pe = -1;
        for (int k = 0; k < np; ++k) {
            int g_, mi_, off_; geo_of(__builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)), row, g_, mi_, off_);
            mn_mi = imin(mn_mi, mi_); mx_mi = imax(mx_mi, mi_); min_pb = imin(min_pb, g_ & 0xfff); max_pe = imax(max_pe, (g_ >> 12) & 0xfff);
        }
        if (np == 0) min_pb = 0;
        set_band(std::false_type{}, mn_mi, mx_mi, min_pb);
        const int Wr = (end_sn - beg_sn + 1) * PN;
        to_ring = Wr <= RC;
        if (DIR && !to_ring) row_spill = true;                      // a row too wide for the score ring: its successors read it from HBM
        if (!reserve()) return 2;
        T *H = io.planes + (long long)off_pn * PN;
        const int my_slot = (row & (RR - 1)) * (NPW * RCS);
        const int nch = (Wr + 63) >> 6;
        refresh_qc();
        const int *mrow = s_mx + base * m1;
        int first = 0, first2 = 0;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // HBM gathers below read cells this wave stored earlier
        for (int c = 0; c < nch; ++c) {
            const int rel = c * 64 + lane, col = beg_sn * PN + rel;
            const bool in_band = rel < Wr;
            int qc = c == 0 ? qoff0 : qoff1;
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? qat(col - 1) : m;
            const int q = mrow[qc];
            int Mv = inf, E1v = inf, E2v = inf, kb = 0, kE1 = 1, kE2 = 1;
            for (int k = 0; k < np; ++k) {
                int g_, mi_, off_; const int p = __builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)); geo_of(p, row, g_, mi_, off_);
                if ((g_ & GEO_RING) && row - p < RR) {
                    if (k == 0) from_ring(0, p, g_, col, Mv, E1v, E2v, kb, 1, kE1, kE2, q); else from_ring(1, p, g_, col, Mv, E1v, E2v, kb, k + 1, kE1, kE2, q);
                } else {
                    const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
                    const int x = col - pb * PN;
                    const bool inH = in_band && (unsigned)x < (unsigned)(Wp + PN), inE = in_band && (unsigned)x < (unsigned)Wp;
                    const T *Hp = io.planes + (long long)(uint32_t)(off_ - (DIR ? (pe - pb + 1) * CWR : 0)) * PN;      // (DIR: its score records, in front of its direction words)
                    int hval = inf, ev1 = inf, ev2 = inf;
                    if (inH && (unsigned)(x - 1) < (unsigned)Wp) hval = gld_cell((GLOBAL_AS const T *)(Hp + (long long)(x - 1) * CWR));
                    if constexpr (GAP == 0) {      // linear gaps: max(H[col-1] + q, H[col] - e) inside the predecessor's vector range, "inf" outside (the first predecessor too)
                        int vert = inf; if (inH && (unsigned)x < (unsigned)Wp) vert = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CWR));
                        const int t = imax(wr(hval + q), wr(vert - e1));
                        Mv = k == 0 ? (inH ? t : inf) : (inH ? imax(Mv, t) : Mv);
                        continue;
                    }
                    if constexpr (CPK) { if (inE) { const int h0 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CWR));
                            const unsigned dd = (unsigned)gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CWR + 1));
                                                    ev1 = (dd & 0xffffu) == 0xffffu ? inf : h0 - (int)(dd & 0xffffu); ev2 = (dd >> 16) == 0xffffu ? inf : h0 - (int)(dd >> 16); } }
                    else
                    if (inE) { ev1 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CWR + PL_E1)); if (GAP == 2) ev2 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CWR + PL_E2)); }
                    if (k == 0) { Mv = hval; E1v = ev1; E2v = ev2; kb = 1; kE1 = 1; kE2 = 1; }
                    else { kb = (inH && hval > Mv) ? k + 1 : kb; if (DIR) { kE1 = (inE && ev1 > E1v) ? k + 1 : kE1; if (GAP == 2) kE2 = (inE && ev2 > E2v) ? k + 1 : kE2; }
                           Mv = inH ? imax(Mv, hval) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v; }
                }
            }
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot, kE1, kE2);
        }
        if (to_ring) pad_ring(nch, my_slot);
        return 1;
    };

#ifdef ABPOA_HIP_PROFILE
    fseg_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int t0 = 0; t0 < gn - 1 && status == 0 && !zstop; t0 += 64) {
#ifdef ABPOA_HIP_ROW_CENSUS
        const long long cen_ts = (long long)__builtin_amdgcn_s_memtime();
#endif
        if (t0 > 0) {       // geometry of the finished tile goes to HBM in one coalesced burst (older predecessors, backtrack, trace)
            const int rb = t0 - 64 + lane; if (wid == 0) { io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN;
                    io.row_max_i[rb] = vg_mi; }
            if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1;
        }
        switch_tile(t0);
#ifdef ABPOA_HIP_ROW_CENSUS
        { asm volatile("" :: "v"(tv_meta), "v"(tv_p0), "v"(tv_tb), "v"(tv_rterm)); fseg[4] += (1ll << 40) + ((long long)__builtin_amdgcn_s_memtime() - cen_ts); }
#endif
        const int r_hi = imin(t0 + 64, gn - 1);
        auto commit_row = [&](int ti, bool ring) __attribute__((always_inline)) {   // v_writelane x3 (no clang builtin); M0 = lane select (two different SGPRs would break the constant-bus limit)
            const int geo_new = sgpr(beg_sn | (end_sn << 12) | (ring ? GEO_RING : 0)), off_new = sgpr(off_pn); mi = sgpr(mi);
            if (__builtin_expect(extend, 0)) {
                const int mx = sgpr(rowmax > inf ? rowmax : inf), rem_row = qlen - rterm + remain_end + 1;      // (a row without a finite cell: max = inf, max_i = -1)
                if (mx > ex_best) { ex_best = mx; ex_i = t0 + ti; ex_j = mi; ex_rem = rem_row; }      // (t0 + ti = the row)
                else if (b.zdrop > 0) { int dd = (ex_rem - rem_row) - (mi - ex_j); if (dd < 0) dd = -dd; if (ex_best - mx > b.zdrop + e1 * dd) zstop = true; }
            }
            if constexpr (WPLAN || !I16) {            // (vg_vm: the all-chunks body centres its int32 arg-max keys on the first predecessor's row maximum)
                const int vm_new = sgpr(rowmax);
                asm volatile("s_mov_b32 m0, %8\n\ts_nop 3\n\tv_writelane_b32 %0, %4, m0\n\tv_writelane_b32 %1, %5, m0\n\tv_writelane_b32 %2, %6, m0\n\tv_writelane_b32 %3, %7, m0"
                             : "+v"(vg_geo), "+v"(vg_mi), "+v"(vg_off), "+v"(vg_vm) : "s"(geo_new), "s"(mi), "s"(off_new), "s"(vm_new), "s"(ti) : "m0");
            } else
            asm volatile("s_mov_b32 m0, %6\n\ts_nop 3\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                         : "+v"(vg_geo), "+v"(vg_mi), "+v"(vg_off) : "s"(geo_new), "s"(mi), "s"(off_new), "s"(ti) : "m0");
        };
        int row = imax(t0, 1);
        while (row < r_hi && !zstop) {
            if constexpr (NW > 1) {
                // ---- NW wavefronts per alignment: the wide body, else wavefront 0 alone with the single-wave bodies below
                const int ti = row & 63;
                last_done = row;
                const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
                rterm = __builtin_amdgcn_readlane(tv_rterm, ti);
                base = meta & 0xff; np = (meta >> 8) & 0xff;
                int rc = 0;
                if ((meta >> 19) & 1) {
                    const int nch_ = ilp_band(row, ti);
                    if (nch_ == -2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                    if (nch_ >= 2) rc = ilp_chunks(std::integral_constant<int, (NCHX + NW - 1) / NW>{}, nch_, row, ti);
                    if (rc != 1) WCOUNT(5);
                } else WCOUNT(1);
                if (rc == 1) { WCOUNT(0); commit_row(ti, true); lds_barrier(); ++row; continue; }      // (barrier: the ring slot is complete before any wavefront reads it)
                rc = 0;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the HBM gathers of the general body read cells other wavefronts stored
                lds_barrier();
                if (wid == 0) {
                    am_key = 0; am_val = INT_MIN; am_v = 0; am_isend = 0; am_any = false;
                    rc = general_body(row, ti);
                    mi = -1;
                    if (rc == 1) {
                        if (I16) {
                            const unsigned kb = wave_max_u32_s(am_key);
                            const int vmax = (int)(kb >> 16) - 32768;
                            rowmax = vmax;
                            if (vmax > inf) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                        } else {
                            const int vmax = wave_max_i32_s(am_any ? am_val : INT_MIN);
                            rowmax = vmax;
                            if (vmax > inf) {
                                unsigned key = 0;
                                if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                                const unsigned kb = wave_max_u32_s(key);
                                mi = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu)) * PN + (PN - 1 - (int)(kb >> 27));
                                if (mi > qlen) mi = -1;
                            }
                        }
                    }
                    if (lane == 0) { bcast[0] = rc; bcast[1] = beg_sn; bcast[2] = end_sn; bcast[3] = off_pn; bcast[4] = mi; bcast[5] = to_ring ? 1 : 0; bcast[6] = cur; bcast[7] = rowmax; }
                }
                lds_barrier();
                rc = __builtin_amdgcn_readfirstlane(bcast[0]); beg_sn = __builtin_amdgcn_readfirstlane(bcast[1]); end_sn = __builtin_amdgcn_readfirstlane(bcast[2]);
                off_pn = __builtin_amdgcn_readfirstlane(bcast[3]); mi = __builtin_amdgcn_readfirstlane(bcast[4]); to_ring = __builtin_amdgcn_readfirstlane(bcast[5]) != 0;
                cur = __builtin_amdgcn_readfirstlane(bcast[6]); rowmax = __builtin_amdgcn_readfirstlane(bcast[7]);
                if (rc == 2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                commit_row(ti, to_ring);
                lds_barrier();                                         // (bcast is free again)
                ++row;
                continue;
            }
            // ---- tight loop over consecutive straight-line rows: only these merge at its back edge (in one loop with the other row
            //      bodies every row paid ~30 register copies for the merge of all paths)
            int ok_ = 1;
            if constexpr (!WPLAN && GAP != 0) if (__builtin_expect(two_chunk_streak, 0)) {      // the band is 65-128 columns wide at the moment: straight to the two-chunk body
                const int ti_ = row & 63, meta_ = __builtin_amdgcn_readlane(tv_meta, ti_);
                if constexpr (DIR) row_spill = (meta_ >> 21) & 1;
                if ((meta_ >> 19) & 1) {
                    CENSUS_T0()
                    last_done = row;
                    rterm = __builtin_amdgcn_readlane(tv_rterm, ti_); base = meta_ & 0xff; np = (meta_ >> 8) & 0xff;
                    const int nch_ = ilp_band(row, ti_);
                    if (nch_ == -2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                    if (nch_ == 2 && ilp_chunks(std::integral_constant<int, 2>{}, nch_, row, ti_) == 1) { commit_row(ti_, true); CENSUS(3) ++row; continue; }
                }
                two_chunk_streak = 0;
            }
            // the tight loop in hand-placed assembly (rows_tight_asm.h): int16, affine, direction words -- the 1 kb production job.  It stops in front of the first
            // row it does not take and says why; that row goes through the bodies below, then the loop is entered again.  (Not in the diagnostic builds, whose
            // ablation bits / clocks / censuses live in the C++ bodies; ABPOA_HIP_DBG bit 11 keeps the compiler's loop, for comparison.)
            bool cxx_tight = true;
#if !defined(ABPOA_HIP_ABLATE) && !defined(ABPOA_HIP_PROFILE) && (!defined(ABPOA_HIP_ROW_CENSUS) || defined(ABPOA_HIP_ASM_CENSUS)) && !defined(ABPOA_HIP_NO_ASM_TIGHT)
            if constexpr (!WPLAN && I16 && GAP == 1 && DIR) {
                if (asm_tight_on && cur + NV * (r_hi - row) <= cap_turbo) {
                    int code, sM, sTB, sRT, sR2, sP0, sM0, sG0, sSL0, sA, sB, sESN, sBSN, sPB0, sPE0, sNV1, sC0, sP1, sM1, sG1, sSL1, sPB1, sPE1, sP2, sSL2, sPB2, sPE2, sP3, sSL3, sPB3, sPE3;
                    long long inb, amok, msk;
#ifdef ABPOA_HIP_ASM_CENSUS
                    const int asm_row0 = row; const long long asm_t0 = (long long)__builtin_amdgcn_s_memtime();
#endif
                    int infw_v = infw, inf_v = inf; asm("" : "+v"(infw_v), "+v"(inf_v));
                    const int mxb = (int)(unsigned)(size_t)(lds_int_t *)s_mx, qb = (int)(unsigned)(size_t)(lds_int_t *)(const void *)s_query;
                    asm volatile(TIGHT_ASM_I16_AFFINE_DIR
                        : [row] "+s"(row), [cur] "+s"(cur), [qcb] "+s"(qc_beg_sn), [code] "=&s"(code), [geo] "+v"(vg_geo), [mi] "+v"(vg_mi), [off] "+v"(vg_off), [qoff0] "+v"(qoff0),
                          [qoff1] "+v"(qoff1), [sM] "=&s"(sM), [sTB] "=&s"(sTB), [sRT] "=&s"(sRT), [sR2] "=&s"(sR2), [sP0] "=&s"(sP0), [sM0] "=&s"(sM0), [sG0] "=&s"(sG0), [sSL0] "=&s"(sSL0), [sA] "=&s"(sA),
                          [sB] "=&s"(sB), [sESN] "=&s"(sESN), [sBSN] "=&s"(sBSN), [sPB0] "=&s"(sPB0), [sPE0] "=&s"(sPE0), [sNV1] "=&s"(sNV1), [sC0] "=&s"(sC0), [sP1] "=&s"(sP1),
                          [sM1] "=&s"(sM1), [sG1] "=&s"(sG1), [sSL1] "=&s"(sSL1), [sPB1] "=&s"(sPB1), [sPE1] "=&s"(sPE1), [sP2] "=&s"(sP2), [sSL2] "=&s"(sSL2), [sPB2] "=&s"(sPB2),
                          [sPE2] "=&s"(sPE2), [sP3] "=&s"(sP3), [sSL3] "=&s"(sSL3), [sPB3] "=&s"(sPB3), [sPE3] "=&s"(sPE3), [inb] "=&s"(inb), [amok] "=&s"(amok), [msk] "=&s"(msk)
                        : [lane] "v"(lane), [tvmeta] "v"(tv_meta), [tvtb] "v"(tv_tb), [tvr1] "v"(tv_r1), [tvr2] "v"(tv_r2), [tvp2] "v"(tv_p2), [tvp3] "v"(tv_p3), [vslot] "v"(vslot), [le1] "v"(le1), [cf1] "v"(cf1), [inj1] "v"(inj1), [kN] "v"(kN),
                          [kE] "v"(kE), [vvl] "v"(vvl), [vl] "v"(l), [infwv] "v"(infw_v), [infv] "v"(inf_v), [c1] "s"(1 - w), [c2] "s"(1 + w), [qlen] "s"(qlen), [rc] "s"(RC), [mxb] "s"(mxb), [fastlo] "s"(fast_lo),
                          [e1] "s"(e1), [oe1] "s"(oe1), [infk] "s"((int)((((unsigned)(inf + 32768)) << 16) | 0xffffu)), [planes] "s"(io.planes), [rhi] "s"(r_hi), [qb] "s"(qb), [m] "s"(m), [perm] "s"(0x05040100)
                        : "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119",
                          "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "vcc", "scc", "m0", "memory");
                    ok_ = code == 1 ? 0 : (code == 2 ? -1 : 1);
                    cxx_tight = false;
#ifdef ABPOA_HIP_ASM_CENSUS
                    fseg[0] += ((long long)(row - asm_row0) << 40) + ((long long)__builtin_amdgcn_s_memtime() - asm_t0);
#endif
                }
            }
#endif
            if constexpr (!WPLAN) if (cxx_tight) for (;;) {
                const int ti_ = row & 63;
                const int meta_ = __builtin_amdgcn_readlane(tv_meta, ti_);
                if (!__builtin_expect((meta_ >> 17) & 1, 1)) break;
                if constexpr (DIR) row_spill = (meta_ >> 21) & 1;
                CENSUS_T0()
                rterm = __builtin_amdgcn_readlane(tv_rterm, ti_);
                base = meta_ & 0xff; np = (meta_ >> 8) & 0xff;
                ok_ = np == 1 ? turbo_body(std::integral_constant<int, 1>{}, std::false_type{}, row, ti_) : turbo_body(std::integral_constant<int, 2>{}, std::false_type{}, row, ti_);
                if (__builtin_expect(ok_ != 1, 0)) break;
                commit_row(ti_, true);
                CENSUS(np == 1 ? 0 : 1)
                if (++row >= r_hi || zstop) break;
            }
            last_done = row - 1;
            if (row >= r_hi || zstop) break;
            const int ti = row & 63;
            last_done = row;
            CENSUS_T0()
            const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
            rterm = __builtin_amdgcn_readlane(tv_rterm, ti);
            base = meta & 0xff; np = (meta >> 8) & 0xff;
            if constexpr (DIR) row_spill = (meta >> 21) & 1;
            if (!WPLAN && ((meta >> 18) & 1)) {                       // three or four predecessors: the straight-line body, outside the tight loop
                if (turbo_body(std::integral_constant<int, 4>{}, std::true_type{}, row, ti) == 1) { commit_row(ti, true); CENSUS(2) ++row; continue; }
            }
            if (!WPLAN && ((meta >> 20) & 1)) {                       // five to eight predecessors
                if (turbo_body(std::integral_constant<int, 8>{}, std::true_type{}, row, ti) == 1) { commit_row(ti, true); CENSUS(2) ++row; continue; }
            }
            // one or two predecessors and the tight loop declined: most often the row's band reaches one
            if (!WPLAN && (((meta >> 17) & 1) ? ok_ == 0 : (DIR && ((meta >> 22) & 1)))) {
                                                                      // vector beyond its predecessors' (every PN-th row of a chain) -- the copies that take those vectors
                const int ok3 = np == 1 ? turbo_body(std::integral_constant<int, 1>{}, std::true_type{}, row, ti) : turbo_body(std::integral_constant<int, 2>{}, std::true_type{}, row, ti);
                if (ok3 == 1) { commit_row(ti, true); CENSUS(np == 1 ? 0 : 1) ++row; continue; }
            }
            if (GAP != 0 && !WPLAN && ((meta >> 19) & 1)) {           // narrow kernel, the straight-line bodies declined (a band of 65-128 columns for a stretch of
                                                                      // rows, a predecessor beyond the score ring, ...): the all-chunks body with two chunks
                const int nch_ = ilp_band(row, ti);
                if (nch_ == -2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                if (nch_ >= 1 && nch_ <= 2 && ilp_chunks(std::integral_constant<int, 2>{}, nch_, row, ti) == 1) { two_chunk_streak = nch_ == 2; commit_row(ti, true); CENSUS(3) ++row; continue; }
            }
            if (WIDEB && ((meta >> 19) & 1)) {                        // wide band: every chunk of the row at once
                const int nch_ = ilp_band(row, ti);
                if (nch_ == -2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                if (nch_ >= 2) {
                    int ok2;
                    // (one body per chunk count from 4 on: a 4-chunk row in the 5-chunk body computes a fifth chunk of lanes nobody keeps -- a fifth of the row's
                    //  vector instructions; 10 kb reads at 5 % error sit at exactly 4 chunks most of the time)
                    if (nch_ <= 3) ok2 = ilp_chunks(std::integral_constant<int, 3>{}, nch_, row, ti);
#ifndef ABPOA_HIP_WIDE_PAIRED_NCH
                    else if (nch_ == 4) ok2 = ilp_chunks(std::integral_constant<int, 4>{}, nch_, row, ti);
                    else if (nch_ == 5) ok2 = ilp_chunks(std::integral_constant<int, 5>{}, nch_, row, ti);
                    else if (nch_ == 6) ok2 = ilp_chunks(std::integral_constant<int, 6>{}, nch_, row, ti);
#else
                    else if (nch_ <= 5) ok2 = ilp_chunks(std::integral_constant<int, 5>{}, nch_, row, ti);
#endif
                    else if (nch_ <= 7) ok2 = ilp_chunks(std::integral_constant<int, 7>{}, nch_, row, ti);
                    else if constexpr (XL) {
                        if (nch_ == 8) ok2 = ilp_chunks(std::integral_constant<int, 8>{}, nch_, row, ti);
                        else if (nch_ == 9) ok2 = ilp_chunks(std::integral_constant<int, 9>{}, nch_, row, ti);
                        else ok2 = ilp_chunks(std::integral_constant<int, 11>{}, nch_, row, ti);
                    }
                    else ok2 = 0;
                    if (ok2 == 1) { WCOUNT(0); commit_row(ti, true); FSTAMP(5) ++row; continue; }
                    WCOUNT(5);
                }
            } else if (WIDEB) WCOUNT(1);
            am_key = 0; am_val = INT_MIN; am_v = 0; am_isend = 0; am_any = false;
            int rc = 0;
            {
                if (!WPLAN && ((meta >> 16) & 1)) {      // (the wide kernels keep only the general body as fall-back: code size)
                    if (np == 1) rc = fast_body(std::integral_constant<int, 1>{}, row, ti);
                    else if (np == 2) rc = fast_body(std::integral_constant<int, 2>{}, row, ti);
                    else rc = fast_body(std::integral_constant<int, 4>{}, row, ti);
                }
                if (rc == 0) rc = general_body(row, ti);
                if (rc == 2) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
                FSTAMP(3)
                // ---- row arg-max (tie-break: lowest lane residue, then the end_sn vector, then the lowest vector), reference :1043-1057
                mi = -1;
                if (I16) {
                    const unsigned kb = wave_max_u32_s(am_key);
                    const int vmax = (int)(kb >> 16) - 32768;
                    rowmax = vmax;
                    if (vmax > inf) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                } else {
                    const int vmax = wave_max_i32_s(am_any ? am_val : INT_MIN);
                    rowmax = vmax;
                    if (vmax > inf) {
                        unsigned key = 0;
                        if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                        const unsigned kb = wave_max_u32_s(key);
                        mi = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu)) * PN + (PN - 1 - (int)(kb >> 27));
                        if (mi > qlen) mi = -1;
                    }
                }
            }
            commit_row(ti, to_ring);
            FSTAMP(4)
            CENSUS(3)
#ifdef ABPOA_HIP_ROW_CENSUS
            fseg[5] += np > 4 ? (1ll << 40) : (((meta >> 16) & 1) ? (1ll << 20) : 1ll);      // why the exact bodies: > 4 predecessors | straight-line body declined | a predecessor beyond the ring
#endif
            ++row;
        }
        if (status != 0) break;
    }
    // ---- geometry of the last (partial) tile
    if (status == 0) {
        const int tb = last_done & ~63, rb = tb + lane;
        if (rb <= last_done) { if (wid == 0) { io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN; io.row_max_i[rb] = vg_mi; }
                               if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1; }
    }
    WG_SYNC();
    // ---- max_pos_left/right as the reference leaves them (only when the caller reads them back)
    if (status == 0 && b.want_lr && wid == 0) {
        const int push_lim = zstop ? last_done : gn;
        for (int r = lane; r < gn; r += 64) {
            int lf = gn, rt = 0;
            if (r == 0) { lf = 0; rt = 0; }
            else for (int k = io.pred_off[r]; k < io.pred_off[r + 1]; ++k) {
                const int p = io.pred_row[k];
                if (p >= push_lim) continue;      // (extension mode, z-drop: the row that ended the loop and the rows behind it pushed nothing, reference :1021-1024)
                const int oi = (p == 0 ? 0 : io.row_max_i[p]) + 1;
                lf = imin(lf, oi); rt = imax(rt, oi);
            }
            io.g_left[r] = lf; io.g_right[r] = rt;
        }
    }
    if (best_out) { best_out[0] = ex_best; best_out[1] = ex_i; best_out[2] = ex_j; }
    cursor_out = (long long)cur * PN; n_cells_out = (long long)__builtin_amdgcn_readlane(wave_scan_add_i32(n_vec_lane), 63) * PN; rows_done_out = last_done;
}

// The fast path is two kernels -- row loop, then global best + backtrack -- so that the row loop's register allocation
// (its SGPR budget above all) is not shared with the tail; the hand-over is the AlnOut record in HBM.
template <typename T, int GAP, int NW = 1, bool WIDEB = false, bool DIR = false, bool XL = false>
__device__ __forceinline__ void align_fast_rows(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    const int lane = threadIdx.x & 63;
    FastIO<T> io;
    io.row_base = vgpr_ptr(b.row_base + d.row0); io.row_remain = vgpr_ptr(b.row_remain + d.row0); io.row_sdist = vgpr_ptr((DIR ? b.row_sdist : b.row_base) + d.row0);
    io.pred_off = vgpr_ptr(b.pred_off + d.poff0); io.pred_row = vgpr_ptr(b.pred_row + d.pred0);
    io.g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0); io.g_esn = vgpr_ptr(b.dp_end_sn + d.row0); io.row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    io.g_left = vgpr_ptr(b.left + d.row0); io.g_right = vgpr_ptr(b.right + d.row0); io.g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    io.planes = (T *)(b.planes + d.plane_off);
    uint8_t *s_query = lds_raw + b.lds.q_off;
    { GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
      if constexpr (NW > 1 || WIDEB) {      // two codes to a byte (rows_fast: qat)
          for (int i = (NW > 1 ? (int)threadIdx.x : lane); 2 * i < d.qlen; i += NW * 64) { const int lo_ = g_query[2 * i], hi_ = 2 * i + 1 < d.qlen ? (int)g_query[2 * i + 1] : 0;
                  s_query[i] = (uint8_t)((lo_ & 15) | (hi_ << 4)); }
      } else for (int i = (NW > 1 ? (int)threadIdx.x : lane); i < d.qlen; i += NW * 64) s_query[i] = g_query[i]; }
    WG_SYNC();
    long long cursor = 0, n_cells = 0; int status = 0, rows_done = 0, last_done = 0;
    const long long clk0 = (long long)__builtin_amdgcn_s_memtime();
    long long fseg[6] = {0, 0, 0, 0, 0, 0};
    int best3[3] = {d.inf_min, 0, 0};
    rows_fast<T, GAP, NW, WIDEB, DIR, XL>(b, d, io, s_query, cursor, n_cells, status, rows_done, last_done, fseg, best3);
    const long long clk1 = (long long)__builtin_amdgcn_s_memtime();
#if !defined(ABPOA_HIP_WIDE_COUNTERS) && !defined(ABPOA_HIP_ROW_CENSUS)
    fseg[5] = (long long)__builtin_amdgcn_s_getreg(63492) | ((long long)__builtin_amdgcn_s_getreg(6164) << 32);      // HW_ID | XCC_ID << 32: where the wave ran (ABPOA_HIP_IMBAL placement report)
#endif
    if (NW > 1 ? threadIdx.x == 0 : lane == 0) { GLOBAL_AS AlnOut *o = vgpr_ptr(out_rec); o->status = status; o->n_cells = n_cells; o->cells_used = cursor; o->clk_dp = clk1 - clk0;
            o->n_rows_done = rows_done; for (int i_ = 0; i_ < 6; ++i_) o->seg[i_] = fseg[i_];
            if (b.align_mode == ABPOA_HIP_EXTEND_MODE) { o->best_score = best3[0]; o->best_row = best3[1]; o->best_col = best3[2]; } }
}


}  // namespace abpoa_hip
