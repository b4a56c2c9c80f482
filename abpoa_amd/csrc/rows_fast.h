#pragma once
#include "dp_common.h"

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
template <typename T, int GAP> struct FastFmt { static constexpr int CW = GAP == 1 ? 4 : 8; };
template <typename T> struct FastIO {
    GLOBAL_AS const uint8_t *row_base; GLOBAL_AS const int32_t *row_remain, *pred_off, *pred_row;
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
#ifdef ABPOA_HIP_ABLATE
#define ABL(BIT) (b.dbg & (BIT))
#else
#define ABL(BIT) false
#endif
template <typename T, int GAP>
__device__ __forceinline__ void rows_fast(const DevBatch &b, const AlnDesc &d, const FastIO<T> &io, const uint8_t *s_query,
                                          long long &cursor_out, long long &n_cells_out, int &status, int &rows_done_out, int &last_done, long long *fseg) {
#ifdef ABPOA_HIP_PROFILE
    long long fseg_last = 0;
#endif
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int CW = FastFmt<T, GAP>::CW;          // values per arena cell record
    constexpr bool I16 = sizeof(T) == 2;
    constexpr int NPW = I16 ? (GAP == 2 ? 2 : 1) : (GAP == 2 ? 3 : 2);
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    constexpr int GEO_RING = 1 << 24;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, m1 = b.m + 1, w = d.w;
    const int inf = d.inf_min;
    const int e1 = b.e1, o1 = b.o1, oe1 = b.o1 + b.e1, e2 = b.e2, o2 = b.o2, oe2 = b.o2 + b.e2;
    const int RR = b.lds.fr_rows, RC = b.lds.fr_cols, RCS = RC + 4;
    int *fr = (int *)(lds_raw + b.lds.phase_off + b.lds.fr_off);
    // LDS byte address of ring row (r & (RR - 1)), column 0, held by lane r & 63: RR divides 64, so one lane-constant VGPR serves
    // every row -- a v_readlane replaces the and / mul / shift / add chain per predecessor and for the row's own slot
    typedef __attribute__((address_space(3))) int lds_int_t;
    const int vslot = (int)(unsigned)(size_t)(lds_int_t *)fr + 4 * ((threadIdx.x & 63 & (RR - 1)) * (NPW * RCS) + 2);
    auto ring_at = [&](int slot_addr, int col_idx) __attribute__((always_inline)) { return (const int *)(lds_int_t *)(size_t)(unsigned)(slot_addr + 4 * col_idx); };
    int *s_mx = (int *)(lds_raw + b.lds.mx_off);
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

    // ---- LDS: extended score matrix (column m = 0) and the score ring, everything "inf"
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < m * m1; i += 64) { const int bb = i / m1, qc = i - bb * m1; s_mx[i] = qc < m ? g_mat[bb * m + qc] : 0; } }
    for (int i = lane; i < RR * NPW * RCS; i += 64) { const int pl = (i / RCS) % NPW; fr[i] = (I16 && pl == 0) ? infw : inf; }
    __syncthreads();
    auto ring_put = [&](int slot, int x, int H, int E1, int E2) __attribute__((always_inline)) {
        int *q = fr + slot * (NPW * RCS) + 2 + x;
        if (I16) { q[0] = (int)(((unsigned)H & 0xffffu) | ((unsigned)E1 << 16)); if (GAP == 2) q[RCS] = E2; }
        else { q[0] = H; q[RCS] = E1; if (GAP == 2) q[2 * RCS] = E2; }
    };

    int cur = 0, n_vec_lane = 0;                    // arena cursor in units of PN cells (one reference SIMD vector); cell count: per-lane sums of the flushed rows' vectors
    const int cap_pn = (int)(d.plane_cap / PN > 0x7fffffffLL ? 0x7fffffffLL : d.plane_cap / PN);
    const int cap_turbo = cap_pn - NV * CW;       // arena room test of the straight-line rows (at most NV vectors)
    const int remain_end = __builtin_amdgcn_readfirstlane(io.row_remain[gn - 1]);
    // ------------------------------------------------------------------ row 0, reference :553-662
    int vg_geo = 0, vg_mi = 0, vg_off = 0;          // lane = row & 63: beg_sn | end_sn << 12 | in-ring << 24, arg-max column, arena offset / PN
    {
        const int r = __builtin_amdgcn_readfirstlane(io.row_remain[0]) - remain_end - 1;
        const int dp_end0 = imin(qlen, imax(0, qlen - r) + w);
        const int end_sn0 = dp_end0 / PN, W0 = (end_sn0 + 1) * PN;
        if ((long long)W0 * CW > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; cursor_out = 0; n_cells_out = 0; rows_done_out = 0; return; }
        const bool ring0 = W0 <= RC;
        T *H = io.planes;
        for (int i = lane; i < W0; i += 64) {
            int h, x1 = inf, x2 = inf, f1 = inf, f2 = inf;
            if (GAP == 1) { const int g = wr(-o1 - e1 * i); h = i == 0 ? 0 : g; x1 = i == 0 ? wr(-oe1) : inf; f1 = i == 0 ? inf : g; }
            else {
                const int g1 = wr(-o1 - e1 * i), g2 = wr(-o2 - e2 * i);
                h = i == 0 ? 0 : imax(g1, g2); x1 = i == 0 ? wr(-oe1) : inf; x2 = i == 0 ? wr(-oe2) : inf; f1 = i == 0 ? inf : g1; f2 = i == 0 ? inf : g2;
            }
            T *cellp = H + (long long)i * CW;
            cellp[0] = (T)h; cellp[PL_E1] = (T)x1; cellp[PL_F1] = (T)f1;
            if (GAP == 2) { cellp[PL_E2] = (T)x2; cellp[PL_F2] = (T)f2; }
            if (ring0) ring_put(0, i, h, x1, x2);
        }
        cur = (end_sn0 + 1) * CW;
        if (lane == 0) { vg_geo = (end_sn0 << 12) | (ring0 ? GEO_RING : 0); vg_mi = 0; vg_off = 0; }     // source: successors get left = right = 1 (:556-561)
    }

    // ------------------------------------------------------------------ static metadata, two tiles ahead
    struct MetaA { int ps, pe, base, rem; };
    struct MetaB { int p[4]; };
    auto load_a = [&](int t0) __attribute__((always_inline)) {
        MetaA a; const int r = imin(t0 + lane, gn - 1);
        a.ps = io.pred_off[r]; a.pe = io.pred_off[r + 1]; a.base = io.row_base[r]; a.rem = io.row_remain[r];
        return a;
    };
    auto load_b = [&](const MetaA &a) __attribute__((always_inline)) {
        MetaB q; const int np = a.pe - a.ps;
#pragma unroll
        for (int k = 0; k < 4; ++k) q.p[k] = io.pred_row[a.ps + imin(k, imax(np - 1, 0))];
        return q;
    };
    MetaA a1 = load_a(0); MetaB b1 = load_b(a1); MetaA a2 = load_a(64);
    int tv_meta = 0, tv_rterm = 0, tv_ps = 0, tv_p0 = 0, tv_p1 = 0, tv_p2 = 0, tv_p3 = 0;
    int tv_tb = 0;          // turbo rows: dist(pred 0) | dist(pred 1) << 8 | (base * (m + 1) * 4) << 16
    auto switch_tile = [&](int t0) __attribute__((always_inline)) {
        const int myrow = t0 + lane, np = a1.pe - a1.ps;
        bool fastrow = np >= 1 && np <= 4 && myrow < gn - 1 && myrow >= 1;
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int dk = myrow - b1.p[k]; fastrow = fastrow && dk >= 1 && dk < RR; }
        tv_meta = (a1.base & 0xff) | (imin(np, 255) << 8) | (fastrow ? (1 << 16) : 0) | ((fastrow && np <= 2 && RC <= 128) ? (1 << 17) : 0) | ((fastrow && np >= 3 && RC <= 128) ? (1 << 18) : 0);      // bit 17: straight-line body (pads 128 ring columns)
        tv_tb = ((myrow - b1.p[0]) & 0xff) | (((myrow - b1.p[1]) & 0xff) << 8) | (((a1.base & 0xff) * m1 * 4) << 16);
        tv_rterm = qlen - (a1.rem - remain_end - 1); tv_ps = a1.ps;
        tv_p0 = b1.p[0]; tv_p1 = b1.p[1]; tv_p2 = b1.p[2]; tv_p3 = b1.p[3];
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
            qoff0 = (c0 >= 1 && c0 <= qlen) ? (int)s_query[c0 - 1] : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? (int)s_query[c1 - 1] : m;
        }
    };
    // one predecessor's contribution from the score ring (k == 0: unmasked, see the header comment)
    // kb: 1 + list index of the first predecessor that supplies the maximum of H[.][col-1] (kidx = this one's 1 + list index): the match flag
    auto from_ring = [&](int k, int p, int g_, int col, int &Mv, int &E1v, int &E2v, int &kb, int kidx) __attribute__((always_inline)) {
        const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
        const int x = col - pb * PN;
        const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), med3i(x - 1, -2, RC));
        int hm1, ev1, ev2 = inf;
        if (I16) { const int w0 = src[0], w1 = src[1]; hm1 = (int)(short)w0; ev1 = w1 >> 16; if (GAP == 2) ev2 = src[RCS + 1]; }
        else { hm1 = src[0]; ev1 = src[RCS + 1]; if (GAP == 2) ev2 = src[2 * RCS + 1]; }
        if (k == 0) { Mv = hm1; E1v = ev1; E2v = ev2; kb = kidx; }
        else {
            const bool inH = (unsigned)x < (unsigned)(Wp + PN), inE = (unsigned)x < (unsigned)Wp;
            kb = (inH && hm1 > Mv) ? kidx : kb;
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        }
    };
    // everything of a chunk after the predecessor gather: F, H, E, stores, ring, arg-max candidate (reference :854-883 / :972-1008)
    auto chunk_tail = [&](int c, int nch, int Wr, int Mv, int E1v, int E2v, int q, int kb, int &first, int &first2, T *H, int my_slot) __attribute__((always_inline)) {
        const int rel = c * 64 + lane, col = beg_sn * PN + rel, vb = beg_sn + c * NV, v = vb + vvl;
        const bool in_band = rel < Wr;
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
        if (ABL(1)) {}
        else if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(H + (long long)rel * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2 & 0xffff; rec.w = mflag; *(int4 *)(H + (long long)rel * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(H + (long long)rel * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(H + (long long)rel * CW); dst[0] = r0; dst[1] = r1; }
        if (to_ring && !ABL(2)) {
            int *qd = fr + my_slot + 2 + rel;
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
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
            qd[0] = infw; if (NPW > 1) qd[RCS] = inf; if (NPW > 2) qd[2 * RCS] = inf;
        }
    };
    // reserve the row's arena cells; false = overflow
    auto reserve = [&]() __attribute__((always_inline)) {
        const int nvr = end_sn - beg_sn + 1;
        if (cur + nvr * CW > cap_pn) return false;
        off_pn = cur; cur += nvr * CW;
        return true;
    };


    // ---- TURBO body: the dominant row shape as straight-line code -- 1 or 2 predecessors (both in the rings), band of at
    //      most 64 columns (one chunk), every vector takes the closed-form F scan (end_sn <= max_pre_end_sn), not the last
    //      query vector, and no value near the wrap limit.  Exactly two rarely-taken exits, both before any side effect.
    //      Returns 1 = done (mi set), 0 = not applicable.
    int mi = -1;
    const int lane4 = lane * 4;
    // arg-max key constants (normal / end_sn vector): lane residue, vector priority, and -- never decisive, it only saves the decoding -- the lane
    const int kN = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)(NV - 1 - vvl) << 8) | (unsigned)lane), kE = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (8u << 8) | (unsigned)lane);
    auto turbo_body = [&](auto npc, int row, int ti) __attribute__((always_inline)) -> int {
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
        if (NPC == 4) {      // three or four predecessors (np at run time); a missing fourth repeats the third
            p2 = __builtin_amdgcn_readlane(tv_p2, ti); p3 = np > 3 ? __builtin_amdgcn_readlane(tv_p3, ti) : p2;
            g2 = __builtin_amdgcn_readlane(vg_geo, p2 & 63); g3 = __builtin_amdgcn_readlane(vg_geo, p3 & 63);
            const int m2_ = __builtin_amdgcn_readlane(vg_mi, p2 & 63), m3_ = __builtin_amdgcn_readlane(vg_mi, p3 & 63);
            mn = sgpr(imin(mn, imin(m2_, m3_))); mx = sgpr(imax(mx, imax(m2_, m3_)));
            min_pb = imin(min_pb, imin(g2 & 0xfff, g3 & 0xfff)); max_pe = imax(max_pe, imax((g2 >> 12) & 0xfff, (g3 >> 12) & 0xfff)); ring &= g2 & g3;
        }
        set_band(std::true_type{}, mn, mx, min_pb);
        const int nvr = end_sn - beg_sn + 1;
        // all conditions as sign bits: (x <= y) <=> (x - y - 1) < 0
        // (arena room: checked for a full-width row, cap_turbo = cap_pn - NV * CW)
        const int okbits = (nvr - NV - 1) & (end_sn - max_pe - 1) & (cur - cap_turbo - 1) & (ring << 7);      // GEO_RING (bit 24) -> bit 31
        if (__builtin_expect(okbits >= 0, 0)) return 0;
        const int Wr = nvr * PN;
        if (__builtin_expect(beg_sn != qc_beg_sn, 0)) {
            qc_beg_sn = beg_sn;
            const int c0 = beg_sn * PN + lane, c1 = c0 + 64;
            qoff0 = (c0 >= 1 && c0 <= qlen) ? (int)s_query[c0 - 1] : m; qoff1 = (c1 >= 1 && c1 <= qlen) ? (int)s_query[c1 - 1] : m;
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
        if (NPC == 4) {
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
        const unsigned rec_off = (unsigned)(cur * (int)(PN * sizeof(T)) + lane * (int)(CW * sizeof(T)));            // arena byte offset of this lane's record (cur = the row's offset once committed)
        asm volatile("" :: "v"(key_c), "v"(qd_addr), "v"(rec_off));      // (materialised here, not sunk to their uses)
        __builtin_amdgcn_sched_barrier(0);
        if (I16) { Mv = (int)(short)raw0; E1v = raw1 >> 16; E2v = raw2; } else { Mv = raw0; E1v = raw1; E2v = raw2; }
        const int Mv_first = Mv;                                   // (match flag below: which predecessor supplies the diagonal)
        int kfirst = 1;                                            // 1 + index of the first predecessor that reaches the running maximum of H[.][col-1]
        auto merge_pred = [&](int r0_, int r1_, int r2_, int x_, int Wp_, int kidx) __attribute__((always_inline)) {
            int hm1, ev1, ev2 = inf;
            if (I16) { hm1 = (int)(short)r0_; ev1 = r1_ >> 16; ev2 = r2_; } else { hm1 = r0_; ev1 = r1_; ev2 = r2_; }
            const bool inH = (unsigned)x_ < (unsigned)(Wp_ + PN), inE = (unsigned)x_ < (unsigned)Wp_;
            if (NPC == 4) kfirst = (inH && hm1 > Mv) ? kidx : kfirst;
            Mv = inH ? imax(Mv, hm1) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v;
        };
        if (NPC >= 2) {
            asm volatile("" : "+v"(rb0), "+v"(rb1));               // (the loads above stay unconditional)
            if (GAP == 2) asm volatile("" : "+v"(rb2));
            merge_pred(rb0, rb1, rb2, x1, Wp1, 2);
        }
        if (NPC == 4) {
            asm volatile("" : "+v"(rc0), "+v"(rc1), "+v"(rd0), "+v"(rd1));
            if (GAP == 2) asm volatile("" : "+v"(rc2), "+v"(rd2));
            merge_pred(rc0, rc1, rc2, x2, Wp2, 3);
            merge_pred(rd0, rd1, rd2, x3, Wp3, 4);                 // (np == 3: the third predecessor again -- no change, kfirst keeps 3 or less)
        }
        const int h = Mv + q;                                      // no wrap possible once the check below passes
        int lowest = imin(h, E1v); if (GAP == 2) lowest = imin(lowest, E2v);
        const bool near_wrap = __any(in_band && lowest < fast_lo);      // decided here, acted on after the scan below: the compare runs beside it, the
                                                                        // branch is off the row's dependent chain (nothing is stored before it)
        int hs = h; if (GAP == 2) hs = imax(imax(h, E1v), E2v);
        // lane 0's scan input is first - e (first = H of the band's first column before any E / F merge = h of lane 0): the shift leaves lane 0's
        // own h - e in place, no trip through an SGPR
        const int g1s = hs + le1;
        int F1 = imax(wave_scan_max_i32(wave_shr1(h - e1, g1s)) - cf1, inj1), F2 = inf;
        if (GAP == 2) { const int g2s = hs + le2; F2 = imax(wave_scan_max_i32(wave_shr1(h - e2, g2s)) - cf2, inj2); }
        if (__builtin_expect(near_wrap, 0)) return 0;
        // ---- from here on the row is committed
        off_pn = cur; cur += nvr * CW;
        int Hout, E1out, E2out = inf;
        if (GAP == 1) {
            const int tmp = imax(h, E1v);
            Hout = imax(tmp, F1);
            E1out = (Hout == tmp) ? imax(E1v - e1, Hout - oe1) : inf;
        } else {
            Hout = imax(hs, imax(F1, F2));
            E1out = imax(E1v - e1, Hout - oe1); E2out = imax(E2v - e2, Hout - oe2);
        }
        // record address = arena base + a 32-bit byte offset (an arena is far below 4 GB): one VALU add, no 64-bit pointer arithmetic per row
        T *const H = (T *)((char *)io.planes + (size_t)rec_off) - lane * CW;
        const int he = I16 ? (int)__builtin_amdgcn_perm((unsigned)E1out, (unsigned)Hout, 0x05040100u) : 0;      // H | E1 << 16 (int16: also the score-ring word)
        // match flag for the backtrack (spare slot of the record, finish_alignment PL_FLAG): 1 + index of the first predecessor k (list order) with
        // H[k][col-1] + q == H[col], 0 = none.  Only a predecessor that supplies the maximum Mv can satisfy it, and only when H == Mv + q.
        // (A predecessor value read from outside its band is `inf`: the backtrack re-checks the column range before it trusts the flag.)
        const int mflag = (h == Hout) ? (NPC == 4 ? kfirst : ((NPC == 2 && Mv != Mv_first) ? 2 : 1)) : 0;
        if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)(H + lane * CW) = rec; }
        else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)(H + lane * CW) = rec; }
        else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; int4 *dst = (int4 *)(H + lane * CW); dst[0] = r0; dst[1] = r1; }
        {
            int *qd = (int *)ring_at(qd_addr, 0);
            if (I16) { qd[0] = in_band ? he : infw; if (GAP == 2) qd[RCS] = in_band ? E2out : inf; }
            else { qd[0] = in_band ? Hout : inf; qd[RCS] = in_band ? E1out : inf; if (GAP == 2) qd[2 * RCS] = in_band ? E2out : inf; }
            qd[64] = infw; if (NPW > 1) qd[RCS + 64] = inf; if (NPW > 2) qd[2 * RCS + 64] = inf;      // (RC <= 128 for these rows: tv_meta bit 17)
        }
        // ---- arg-max, reference :1043-1057: value, then lowest lane residue, then the end_sn vector, then the lowest vector
        if (I16) {
            const unsigned key = ((unsigned)Hout << 16) + (unsigned)key_c;
            // columns past the query end exist in the last query vector only: computed and stored like the others, never the row's arg-max (ref :1049-1056)
            const unsigned kb = wave_max_u32_s((in_band && colrel <= qlen) ? key : 0u);
            mi = ((int)(kb >> 16) - 32768 > inf) ? beg_sn * PN + (int)(kb & 63) : -1;      // the winning lane IS the column offset
        } else {
            const bool am_ok = in_band && colrel <= qlen;
            const int vmax = wave_max_i32_s(am_ok ? Hout : INT_MIN);
            const unsigned key = (am_ok && Hout == vmax) ? (((unsigned)(PN - 1 - l) << 12) | (unsigned)((vvl == nvr - 1) ? 8 : NV - 1 - vvl)) : 0u;
            const unsigned kb = wave_max_u32_s(key);
            const int vrel = (kb & 8) ? nvr - 1 : NV - 1 - (int)(kb & 7);
            mi = (vmax > inf) ? (beg_sn + vrel) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)) : -1;
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
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? (int)s_query[col - 1] : m;
            const int q = mrow[qc];
            int Mv = lane, E1v = inf, E2v = inf, kb = 0;
            if (!ABL(16)) from_ring(0, pr[0], pgeo[0], col, Mv, E1v, E2v, kb, 1);
            if (NPC >= 2 && !ABL(16)) from_ring(1, pr[1], pgeo[1], col, Mv, E1v, E2v, kb, 2);
            if (NPC >= 4) { if (np > 2) from_ring(2, pr[2], pgeo[2], col, Mv, E1v, E2v, kb, 3); if (np > 3) from_ring(3, pr[3], pgeo[3], col, Mv, E1v, E2v, kb, 4); }
            FSTAMP(1)
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot);
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
        if (!reserve()) return 2;
        to_ring = Wr <= RC;
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
            if (c >= 2) qc = (col >= 1 && col <= qlen) ? (int)s_query[col - 1] : m;
            const int q = mrow[qc];
            int Mv = inf, E1v = inf, E2v = inf, kb = 0;
            for (int k = 0; k < np; ++k) {
                int g_, mi_, off_; const int p = __builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)); geo_of(p, row, g_, mi_, off_);
                if ((g_ & GEO_RING) && row - p < RR) {
                    if (k == 0) from_ring(0, p, g_, col, Mv, E1v, E2v, kb, 1); else from_ring(1, p, g_, col, Mv, E1v, E2v, kb, k + 1);
                } else {
                    const int pb = g_ & 0xfff, pe = (g_ >> 12) & 0xfff, Wp = (pe - pb + 1) * PN;
                    const int x = col - pb * PN;
                    const bool inH = in_band && (unsigned)x < (unsigned)(Wp + PN), inE = in_band && (unsigned)x < (unsigned)Wp;
                    const T *Hp = io.planes + (long long)(uint32_t)off_ * PN;
                    int hval = inf, ev1 = inf, ev2 = inf;
                    if (inH && (unsigned)(x - 1) < (unsigned)Wp) hval = gld_cell((GLOBAL_AS const T *)(Hp + (long long)(x - 1) * CW));
                    if (inE) { ev1 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CW + PL_E1)); if (GAP == 2) ev2 = gld_cell((GLOBAL_AS const T *)(Hp + (long long)x * CW + PL_E2)); }
                    if (k == 0) { Mv = hval; E1v = ev1; E2v = ev2; kb = 1; }
                    else { kb = (inH && hval > Mv) ? k + 1 : kb; Mv = inH ? imax(Mv, hval) : Mv; E1v = inE ? imax(E1v, ev1) : E1v; if (GAP == 2) E2v = inE ? imax(E2v, ev2) : E2v; }
                }
            }
            chunk_tail(c, nch, Wr, Mv, E1v, E2v, q, kb, first, first2, H, my_slot);
        }
        if (to_ring) pad_ring(nch, my_slot);
        return 1;
    };

#ifdef ABPOA_HIP_PROFILE
    fseg_last = (long long)__builtin_amdgcn_s_memtime();
#endif
    for (int t0 = 0; t0 < gn - 1 && status == 0; t0 += 64) {
        if (t0 > 0) {       // geometry of the finished tile goes to HBM in one coalesced burst (older predecessors, backtrack, trace)
            const int rb = t0 - 64 + lane; io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN; io.row_max_i[rb] = vg_mi;
            if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1;
        }
        switch_tile(t0);
        const int r_hi = imin(t0 + 64, gn - 1);
        auto commit_row = [&](int ti, bool ring) __attribute__((always_inline)) {   // v_writelane x3 (no clang builtin); M0 = lane select (two different SGPRs would break the constant-bus limit)
            const int geo_new = sgpr(beg_sn | (end_sn << 12) | (ring ? GEO_RING : 0)), off_new = sgpr(off_pn); mi = sgpr(mi);
            asm volatile("s_mov_b32 m0, %6\n\ts_nop 3\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %4, m0\n\tv_writelane_b32 %2, %5, m0"
                         : "+v"(vg_geo), "+v"(vg_mi), "+v"(vg_off) : "s"(geo_new), "s"(mi), "s"(off_new), "s"(ti) : "m0");
        };
        int row = imax(t0, 1);
        while (row < r_hi) {
            // ---- tight loop over consecutive straight-line rows: only these merge at its back edge (in one loop with the other row
            //      bodies every row paid ~30 register copies for the merge of all paths)
            int ok_ = 0;
            for (;;) {
                const int ti_ = row & 63;
                const int meta_ = __builtin_amdgcn_readlane(tv_meta, ti_);
                if (!__builtin_expect((meta_ >> 17) & 1, 1)) break;
                rterm = __builtin_amdgcn_readlane(tv_rterm, ti_);
                base = meta_ & 0xff; np = (meta_ >> 8) & 0xff;
                ok_ = np == 1 ? turbo_body(std::integral_constant<int, 1>{}, row, ti_) : turbo_body(std::integral_constant<int, 2>{}, row, ti_);
                if (__builtin_expect(ok_ != 1, 0)) break;
                commit_row(ti_, true);
                if (++row >= r_hi) break;
            }
            last_done = row - 1;
            if (row >= r_hi) break;
            const int ti = row & 63;
            last_done = row;
            const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
            rterm = __builtin_amdgcn_readlane(tv_rterm, ti);
            base = meta & 0xff; np = (meta >> 8) & 0xff;
            if ((meta >> 18) & 1) {                                   // three or four predecessors: the straight-line body, outside the tight loop
                if (turbo_body(std::integral_constant<int, 4>{}, row, ti)) { commit_row(ti, true); ++row; continue; }
            }
            am_key = 0; am_val = INT_MIN; am_v = 0; am_isend = 0; am_any = false;
            int rc = 0;
            {
                if ((meta >> 16) & 1) {
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
                    if (vmax > inf) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                } else {
                    const int vmax = wave_max_i32_s(am_any ? am_val : INT_MIN);
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
            ++row;
        }
        if (status != 0) break;
    }
    // ---- geometry of the last (partial) tile
    if (status == 0) {
        const int tb = last_done & ~63, rb = tb + lane;
        if (rb <= last_done) { io.g_bsn[rb] = vg_geo & 0xfff; io.g_esn[rb] = (vg_geo >> 12) & 0xfff; io.g_coff[rb] = (long long)(uint32_t)vg_off * PN; io.row_max_i[rb] = vg_mi;
                               if (rb >= 1) n_vec_lane += ((vg_geo >> 12) & 0xfff) - (vg_geo & 0xfff) + 1; }
    }
    __syncthreads();
    // ---- max_pos_left/right as the reference leaves them (only when the caller reads them back)
    if (status == 0 && b.want_lr) {
        for (int r = lane; r < gn; r += 64) {
            int lf = gn, rt = 0;
            if (r == 0) { lf = 0; rt = 0; }
            else for (int k = io.pred_off[r]; k < io.pred_off[r + 1]; ++k) {
                const int p = io.pred_row[k]; const int oi = (p == 0 ? 0 : io.row_max_i[p]) + 1;
                lf = imin(lf, oi); rt = imax(rt, oi);
            }
            io.g_left[r] = lf; io.g_right[r] = rt;
        }
    }
    cursor_out = (long long)cur * PN; n_cells_out = (long long)__builtin_amdgcn_readlane(wave_scan_add_i32(n_vec_lane), 63) * PN; rows_done_out = last_done;
}

}  // namespace abpoa_hip
