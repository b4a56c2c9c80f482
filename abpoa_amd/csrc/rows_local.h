// Row loop for LOCAL alignment without a band (reference -m 1: abpoa_post_set_para turns the band off, src/abpoa_align.c:150): every row spans the
// whole query, columns [0, (qlen / pn + 1) * pn), so there is no band to derive and no row-to-row scalar dependency besides the scores
// themselves.  One wavefront per alignment keeps EVERY 64-column chunk of a row in registers at once (the amino-acid workload of BASELINE.json
// configs[4]: 500 residues = 8-9 chunks of int16 cells): the chunks are independent instruction streams that the scheduler interleaves, F is the
// closed form of rows_fast.h (per-chunk prefix maxima + a carry chain over the chunk totals; set_num == pn for every vector in local mode,
// reference :862-868 / :980-986 with `local`), and the row maximum -- what local mode keeps per row, reference :1012-1016 / :1108-1110 -- is one
// packed-key reduction taken from max(0, M + q, E): an F term is some H of the same row minus at least o + e.
//
// Local-mode arithmetic (oracle/abpoa_dp_oracle.c dp_row, reference :781-885 / :887-1010 with the `local` branches):
//   row 0: every plane 0;   M[j] = max_p H_p[j-1] with H_p[-1] = 0 (the reference shifts zeros in);   E[j] = max_p E_p[j]  (no range masks: every
//   row has the same geometry);   H = max(0, M + q, E.., F..);   affine E' = (H == max(M + q, E)) ? max(E - e, H - oe) : 0;
//   convex Ex' = max(0, Ex - ex, H - oex);   F is stored as computed (not clamped).
// Arena: the cell records of the fast row loops (rows_fast.h FastFmt), every row (nvr * pn) records, row r at r * nvr * CW vectors -- the tail
// kernel (backtrack.h) reads them as it reads the banded loops' arenas.
#pragma once
#include "rows_fast.h"

namespace abpoa_hip {

// (alignments this loop takes; must agree between the kernels and the host: engine.cpp)
__device__ __forceinline__ bool takes_local(const DevBatch &b, const AlnDesc &d) {
    // (int16 scores only: a local alignment of at most loc_cols columns cannot need more unless the matrix is extreme -- then the general kernel takes it)
    return b.align_mode == ABPOA_HIP_LOCAL_MODE && b.wb < 0 && b.gap_mode != ABPOA_HIP_LINEAR_GAP && (d.flags & ALN_FAST_OK) && b.lds.loc_cols > 0 && d.bits == 16 &&
           (d.qlen / 16 + 1) * 16 <= b.lds.loc_cols && d.qlen <= b.lds.q_cap && !(b.dbg & 64);
}

template <typename T, int GAP, int NCH>
__device__ __forceinline__ void rows_local(const DevBatch &b, const AlnDesc &d, const FastIO<T> &io, const uint8_t *s_query, AlnOut *out_rec) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int CW = FastFmt<T, GAP>::CW;
    constexpr bool I16 = sizeof(T) == 2;
    constexpr int NPW = I16 ? (GAP == 2 ? 2 : 1) : (GAP == 2 ? 3 : 2);
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, m1 = b.m + 1;
    const int inf = d.inf_min;
    const int e1 = b.e1, o1 = b.o1, oe1 = b.o1 + b.e1, e2 = b.e2, o2 = b.o2, oe2 = b.o2 + b.e2;
    const int RR = b.lds.loc_rows, RC = b.lds.loc_cols, RCS = RC + 4;
    int *fr = (int *)(lds_raw + b.lds.phase_off + b.lds.fr_off);
    int *s_mx = (int *)(lds_raw + b.lds.mx_off);
    typedef __attribute__((address_space(3))) int lds_int_t;
    const int vslot = (int)(unsigned)(size_t)(lds_int_t *)fr + 4 * ((lane & (RR - 1)) * (NPW * RCS) + 2);      // LDS byte address of ring row (lane & (RR - 1)), column 0
    auto ring_at = [&](int slot_addr, int col_idx) __attribute__((always_inline)) { return (int *)(lds_int_t *)(size_t)(unsigned)(slot_addr + 4 * col_idx); };
    auto wr = [](int x) __attribute__((always_inline)) { return (int)(T)x; };
    const int end_sn = qlen / PN, nvr = end_sn + 1, W = nvr * PN, nch = (W + 63) >> 6;
    const int qlen_sn = end_sn;

    // per-lane constants of the F scan and of the arg-max key (as rows_fast.h)
    const int idist = inj_dist<PN>(l);
    const int inj1 = idist >= 0 ? inf - idist * e1 : INT_MIN, inj2 = idist >= 0 ? inf - idist * e2 : INT_MIN;
    const int le1 = lane * e1, le2 = lane * e2, cf1 = oe1 - e1 + le1, cf2 = oe2 - e2 + le2;
    const int kconst = I16 ? (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (unsigned)(2047 - vvl)) : 0;
    const int ktie = ((PN - 1 - l) << 8) | (127 - vvl);          // int32 key: value << 12 | lane residue << 8 | end vector << 7 | vector order (7 bits: up to 9 x 8 vectors)

    long long status_cells = 0; int status = 0;
    if ((long long)gn * W * CW > d.plane_cap) status = ABPOA_HIP_STATUS_OVERFLOW;

    // ---- LDS: extended score matrix (column m = 0) and the score ring: "inf" everywhere, 0 in the H guard cell left of column 0
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < m * m1; i += 64) { const int bb = i / m1, qc = i - bb * m1; s_mx[i] = qc < m ? g_mat[bb * m + qc] : 0; } }
    for (int i = lane; i < RR * NPW * RCS; i += 64) {
        const int pl = (i / RCS) % NPW, x = i % RCS - 2;
        const int hz = I16 ? (int)((unsigned)inf << 16) : 0;          // H = 0 (int16: packed with E1 = inf, which nobody reads there)
        const int infw = I16 ? (int)(((unsigned)inf & 0xffffu) | ((unsigned)inf << 16)) : inf;
        fr[i] = (pl == 0 && x == -1) ? hz : ((I16 && pl == 0) ? infw : inf);
    }
    __syncthreads();

    T *const planes = io.planes + (long long)lane * CW;
    const int row_stride = nvr * PN * CW;                         // values per row in the arena
    // ---- row 0: every plane 0 (reference :553-662 local branch)
    if (status == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) if (c < nch) {
            T *H = planes + c * 64 * CW;
            if (I16 && GAP == 1) { *(int2 *)H = make_int2(0, 0); }
            else if (I16 || GAP == 1) { *(int4 *)H = make_int4(0, 0, 0, 0); }
            else { ((int4 *)H)[0] = make_int4(0, 0, 0, 0); ((int4 *)H)[1] = make_int4(0, 0, 0, 0); }
            int *qd0 = fr + 2 + c * 64 + lane;                       // ring slot 0
            qd0[0] = 0; if (NPW > 1) qd0[RCS] = 0; if (NPW > 2) qd0[2 * RCS] = 0;
        }
    }
    // ---- static metadata, two tiles ahead (as rows_fast.h)
    struct MetaA { int ps, pe, base; };
    struct MetaB { int p[8]; };
    auto load_a = [&](int t0) __attribute__((always_inline)) { MetaA a; const int r = imin(t0 + lane, gn - 1); a.ps = io.pred_off[r]; a.pe = io.pred_off[r + 1]; a.base = io.row_base[r]; return a; };
    auto load_b = [&](const MetaA &a) __attribute__((always_inline)) {
        MetaB q; const int np = a.pe - a.ps;
#pragma unroll
        for (int k = 0; k < 8; ++k) q.p[k] = io.pred_row[a.ps + imin(k, imax(np - 1, 0))];
        return q;
    };
    MetaA a1 = load_a(0); MetaB b1 = load_b(a1); MetaA a2 = load_a(64);
    int tv_meta = 0, tv_ps = 0, tv_p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int vg_mi = 0;                                                  // lane = row & 63: the row's arg-max column (trace / tests)
    int best_score = inf, best_i = 0, best_j = 0, n_rows_done = 0;
    // query codes of this lane's column in every chunk: the geometry never changes, so they are read once
    int qoff[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) { const int col = c * 64 + lane; qoff[c] = (col >= 1 && col <= qlen) ? (int)s_query[col - 1] : m; }

    for (int t0 = 0; t0 < gn - 1 && status == 0; t0 += 64) {
        if (t0 > 0) { const int rb = t0 - 64 + lane; io.g_bsn[rb] = 0; io.g_esn[rb] = end_sn; io.g_coff[rb] = (long long)rb * row_stride; io.row_max_i[rb] = vg_mi; }
        {   // switch tile
            const int np_ = a1.pe - a1.ps;
            tv_meta = (a1.base & 0xff) | (imin(np_, 255) << 8); tv_ps = a1.ps;
#pragma unroll
            for (int k = 0; k < 8; ++k) tv_p[k] = b1.p[k];
            a1 = a2; b1 = load_b(a1); a2 = load_a(t0 + 128);
        }
        const int r_hi = imin(t0 + 64, gn - 1);
        for (int row = imax(t0, 1); row < r_hi; ++row) {
            const int ti = row & 63;
            const int meta = __builtin_amdgcn_readlane(tv_meta, ti), base = meta & 0xff, np = (meta >> 8) & 0xff;
            const int *mrow = s_mx + base * m1;
            int q[NCH], Mv[NCH], E1v[NCH], E2v[NCH], kb[NCH];
#pragma unroll
            for (int c = 0; c < NCH; ++c) q[c] = mrow[qoff[c]];
            // ---- predecessor gather: ring rows (all rows share one geometry: no masks, no clamps) or, for a predecessor older than the ring, the arena
            auto read_pred = [&](int p, int *hc, int *ec1, int *ec2) __attribute__((always_inline)) {
                if (row - p < RR) {
                    const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), lane - 1);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        if (I16) { const int w0 = src[c * 64], w1 = src[c * 64 + 1]; hc[c] = (int)(short)w0; ec1[c] = w1 >> 16; ec2[c] = GAP == 2 ? src[RCS + c * 64 + 1] : inf; }
                        else { hc[c] = src[c * 64]; ec1[c] = src[RCS + c * 64 + 1]; ec2[c] = GAP == 2 ? src[2 * RCS + c * 64 + 1] : inf; }
                    }
                } else {
                    const T *Hp = io.planes + (long long)p * row_stride;
                    gld_wait();                                      // (this wave's earlier score-plane stores are complete)
#pragma unroll
                    for (int c = 0; c < NCH; ++c) {
                        const int x = c * 64 + lane, xh = med3i(x - 1, 0, W - 1), xe = imin(x, W - 1);
                        gld_async_cell(hc[c], Hp + (long long)xh * CW); gld_async_cell(ec1[c], Hp + (long long)xe * CW + PL_E1);
                        if (GAP == 2) gld_async_cell(ec2[c], Hp + (long long)xe * CW + PL_E2); else ec2[c] = inf;
                    }
#pragma unroll
                    for (int c = 0; c < NCH; ++c) { if (GAP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]), "+v"(ec2[c]) :: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]) :: "memory"); }
                    if (lane == 0) hc[0] = 0;                        // H_p[-1] = 0
                }
            };
            {
                int hc[NCH], ec1[NCH], ec2[NCH];
                read_pred(__builtin_amdgcn_readlane(tv_p[0], ti), hc, ec1, ec2);
#pragma unroll
                for (int c = 0; c < NCH; ++c) { Mv[c] = hc[c]; E1v[c] = ec1[c]; E2v[c] = ec2[c]; kb[c] = 1; }
                auto another = [&](int p, int kidx) __attribute__((always_inline)) {
                    int hd[NCH], ed1[NCH], ed2[NCH];
                    read_pred(p, hd, ed1, ed2);
#pragma unroll
                    for (int c = 0; c < NCH; ++c) { kb[c] = hd[c] > Mv[c] ? kidx : kb[c]; Mv[c] = imax(Mv[c], hd[c]); E1v[c] = imax(E1v[c], ed1[c]); if (GAP == 2) E2v[c] = imax(E2v[c], ed2[c]); }
                };
                if (np > 1) { another(__builtin_amdgcn_readlane(tv_p[1], ti), 2); if (np > 2) { another(__builtin_amdgcn_readlane(tv_p[2], ti), 3);
                        if (np > 3) { another(__builtin_amdgcn_readlane(tv_p[3], ti), 4);
                    if (np > 4) { another(__builtin_amdgcn_readlane(tv_p[4], ti), 5); if (np > 5) { another(__builtin_amdgcn_readlane(tv_p[5], ti), 6);
                            if (np > 6) { another(__builtin_amdgcn_readlane(tv_p[6], ti), 7);
                    if (np > 7) { another(__builtin_amdgcn_readlane(tv_p[7], ti), 8);
                        const int ps = __builtin_amdgcn_readlane(tv_ps, ti);
                        for (int k = 8; k < np; ++k) another(__builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + ps + k)), imin(k + 1, 65)); } } } } } } }
            }
            // ---- H before F; per-chunk unseeded prefix maxima of g = hs + lane * e; arg-max key from max(0, M + q, E)
            int h[NCH], hs[NCH], hsE[NCH], g1[NCH], g2[NCH], s1[NCH], s2[NCH];
            unsigned amk = 0;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                h[c] = wr(Mv[c] + q[c]);
                hs[c] = h[c]; if (GAP == 2) hs[c] = imax(imax(h[c], E1v[c]), E2v[c]);
                hsE[c] = GAP == 1 ? imax(h[c], E1v[c]) : hs[c];
                g1[c] = hs[c] + le1; s1[c] = wave_shr1(INT_MIN, g1[c]);
                if (GAP == 2) { g2[c] = hs[c] + le2; s2[c] = wave_shr1(INT_MIN, g2[c]); }
                const int col = c * 64 + lane, vb = c * NV;
                const bool in_band = col < W, is_end = (vb + vvl == end_sn);
                const int cand = (is_end && col > qlen) ? inf : imax(0, hsE[c]);
                unsigned key;
                if (I16) key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
                else key = ((unsigned)imin(imax(cand, 0), 0xFFFFF) << 12) | (unsigned)(ktie - vb) | (is_end ? 128u : 0u);      // (local scores are >= 0; 20 bits)
                amk = (in_band && key > amk) ? key : amk;
            }
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
            const unsigned kbst = (unsigned)__builtin_amdgcn_readlane((int)amk, 63);
            // ---- carry chain over the chunk totals: seed[0] = first - e (first = H of column 0 before E / F), seed[c + 1] = max(total[c], seed[c]) - 64 e
            int seed1[NCH], seed2[NCH];
            seed1[0] = __builtin_amdgcn_readlane(h[0], 0) - e1; seed2[0] = seed1[0] + e1 - e2;
            asm("" : "+v"(seed1[0])); if (GAP == 2) asm("" : "+v"(seed2[0]));
#pragma unroll
            for (int c = 0; c + 1 < NCH; ++c) {
                seed1[c + 1] = imax(__builtin_amdgcn_readlane(imax(s1[c], g1[c]), 63), seed1[c]) - 64 * e1;
                if (GAP == 2) seed2[c + 1] = imax(__builtin_amdgcn_readlane(imax(s2[c], g2[c]), 63), seed2[c]) - 64 * e2;
            }
            // ---- F, H, E of every chunk; records to the arena, H / E to the ring
            T *const Hrow = planes + (long long)row * row_stride;
            int *const qd = ring_at(__builtin_amdgcn_readlane(vslot, ti), lane);
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const int F1 = imax(imax(s1[c], seed1[c]) - cf1, inj1);
                int F2 = inf; if (GAP == 2) F2 = imax(imax(s2[c], seed2[c]) - cf2, inj2);
                int Hout, E1out, E2out = inf;
                if (GAP == 1) {
                    Hout = imax(0, imax(hsE[c], F1));
                    const int en_ = imax(wr(E1v[c] - e1), wr(Hout - oe1));
                    E1out = (Hout == hsE[c]) ? en_ : 0;
                } else {
                    Hout = imax(0, imax(hs[c], imax(F1, F2)));
                    E1out = imax(0, imax(wr(E1v[c] - e1), wr(Hout - oe1)));
                    E2out = imax(0, imax(wr(E2v[c] - e2), wr(Hout - oe2)));
                }
                // match flag for the backtrack (rows_fast.h); 0 where H is 0: the local walk stops there (reference :126), the tail's full step sees it
                const int mflag = (Mv[c] + q[c] == Hout && kb[c] <= 64 && Hout != 0) ? kb[c] : 0;
                const int he = (int)(((unsigned)Hout & 0xffffu) | ((unsigned)E1out << 16));
                if (c < nch) {      // (lanes past the row's end write cells the next row overwrites: same wave, program order; 64 records of slack)
                    T *H = Hrow + c * 64 * CW;
                    if (I16 && GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)H = rec; }
                    else if (I16) { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2 & 0xffff; rec.w = mflag; *(int4 *)H = rec; }
                    else if (GAP == 1) { int4 rec; rec.x = Hout; rec.y = E1out; rec.z = F1; rec.w = mflag; *(int4 *)H = rec; }
                    else { int4 r0, r1; r0.x = Hout; r0.y = E1out; r0.z = E2out; r0.w = F1; r1.x = F2; r1.y = mflag; r1.z = 0; r1.w = 0; ((int4 *)H)[0] = r0; ((int4 *)H)[1] = r1; }
                    if (I16) { qd[c * 64] = he; if (GAP == 2) qd[RCS + c * 64] = E2out; }
                    else { qd[c * 64] = Hout; qd[RCS + c * 64] = E1out; if (GAP == 2) qd[2 * RCS + c * 64] = E2out; }
                }
            }
            // ---- row maximum -> best cell (local: strictly greater, the first row wins ties; reference :1012-1016)
            int rowmax, mi = -1;
            if (I16) { rowmax = (int)(kbst >> 16) - 32768; if (rowmax > inf) { mi = (2047 - (int)(kbst & 0x7ff)) * PN + (PN - 1 - (int)((kbst >> 12) & 0xf)); if (mi > qlen) mi = -1; } }
            else { rowmax = (int)(kbst >> 12); mi = (127 - (int)(kbst & 127)) * PN + (PN - 1 - (int)((kbst >> 8) & 0xf)); if (mi > qlen) mi = -1; }
            if (rowmax > best_score) { best_score = rowmax; best_i = row; best_j = mi; }
            { const int mi_s = sgpr(mi); asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(vg_mi) : "s"(mi_s), "s"(ti) : "m0"); }
            n_rows_done = row;
        }
    }
    if (status == 0) {
        const int tb = n_rows_done & ~63, rb = tb + lane;
        if (rb <= n_rows_done) { io.g_bsn[rb] = 0; io.g_esn[rb] = end_sn; io.g_coff[rb] = (long long)rb * row_stride; io.row_max_i[rb] = vg_mi; }
    }
    if (lane == 0) {
        GLOBAL_AS AlnOut *o = vgpr_ptr(out_rec);
        o->status = status; o->n_cells = status == 0 ? (long long)imax(0, gn - 2) * W : 0; o->cells_used = status == 0 ? (long long)(gn - 1) * row_stride : 0;
        o->clk_dp = 0; o->n_rows_done = n_rows_done; o->best_score = best_score; o->best_row = best_i; o->best_col = best_j;
        for (int i_ = 0; i_ < 6; ++i_) o->seg[i_] = 0;
    }
    (void)status_cells; (void)o1; (void)o2; (void)qlen_sn;
}


// ---------------------------------------------------------------------------------------------------------------------------------------------
// TEAM of NW wavefronts on one local alignment (the workload of BASELINE.json configs[4] as it is defined: 1000 read-sets, ONE alignment each at a time --
// a single wavefront per alignment leaves every SIMD with one wavefront, which issues at most one instruction in four cycles: 76 % of the cycles busy,
// 25 % of the integer-VALU peak, 4 us per 512-column row).  Local mode has no band to derive, so -- unlike the banded loops, where teams lost because
// every wavefront repeats ~370 instructions of per-row bookkeeping -- a row here IS its chunks: wavefront w takes a contiguous share of
// them (the same share in every row: the geometry never changes), and a row needs
//   * ONE exchange through LDS in its middle: each wavefront's carry-chain result as if nothing came in (the chain is max-plus: the incoming seed is
//     folded in afterwards), and its arg-max key;
//   * one barrier at its end, which publishes the row's ring slot to the wavefronts that read it as a predecessor (the diagonal cell at a chunk boundary
//     belongs to the neighbour wavefront).
// Same cells, same records, same best cell as rows_local: the kernels are interchangeable per launch (dp_local_rows.hip picks by launch size).
// NCW: chunks a wavefront keeps in registers (ceil(chunks of the row / NW)).
template <typename T, int GAP, int NCW, int NW>
__device__ __forceinline__ void rows_local_team(const DevBatch &b, const AlnDesc &d, const FastIO<T> &io, const uint8_t *s_query, AlnOut *out_rec) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int CW = FastFmt<T, GAP>::CW;
    constexpr bool I16 = sizeof(T) == 2;
    static_assert(I16, "the local row loops are int16 (takes_local)");
    constexpr int NPW = GAP == 2 ? 2 : 1;
    constexpr int PL_E1 = 1, PL_E2 = 2;
    constexpr int NT = NW * 64;
    const int tid = threadIdx.x, lane = tid & 63, l = lane % PN, vvl = lane / PN;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, m1 = b.m + 1;
    const int inf = d.inf_min;
    const int e1 = b.e1, oe1 = b.o1 + b.e1, e2 = b.e2, oe2 = b.o2 + b.e2;
    const int RR = b.lds.loc_rows, RC = b.lds.loc_cols, RCS = RC + 4;
    int *fr = (int *)(lds_raw + b.lds.phase_off + b.lds.fr_off);
    int4 *xch = (int4 *)(fr + ((RR * NPW * RCS + 3) & ~3));          // exchange slots behind the ring: two parities x NW entries (LdsPlan.total_local reserves them)
    int *s_mx = (int *)(lds_raw + b.lds.mx_off);
    typedef __attribute__((address_space(3))) int lds_int_t;
    const int vslot = (int)(unsigned)(size_t)(lds_int_t *)fr + 4 * ((lane & (RR - 1)) * (NPW * RCS) + 2);
    auto ring_at = [&](int slot_addr, int col_idx) __attribute__((always_inline)) { return (int *)(lds_int_t *)(size_t)(unsigned)(slot_addr + 4 * col_idx); };
    auto wr = [](int x) __attribute__((always_inline)) { return (int)(T)x; };
    const int end_sn = qlen / PN, nvr = end_sn + 1, W = nvr * PN, nch = (W + 63) >> 6;
    // this wavefront's chunks: c0 .. c0 + cnt - 1 (nch / NW each, the first nch % NW wavefronts one more)
    const int bs_ = nch / NW, rm_ = nch - bs_ * NW, cnt = bs_ + (wid < rm_ ? 1 : 0), c0 = wid * bs_ + imin(wid, rm_);

    const int idist = inj_dist<PN>(l);
    const int inj1 = idist >= 0 ? inf - idist * e1 : INT_MIN, inj2 = idist >= 0 ? inf - idist * e2 : INT_MIN;
    const int le1 = lane * e1, le2 = lane * e2, cf1 = oe1 - e1 + le1, cf2 = oe2 - e2 + le2;
    const int kconst = (int)(0x80000000u | ((unsigned)(PN - 1 - l) << 12) | (unsigned)(2047 - vvl));

    int status = 0;
    if ((long long)gn * W * CW > d.plane_cap) status = ABPOA_HIP_STATUS_OVERFLOW;

    // ---- LDS: extended score matrix, score ring ("inf", 0 in the H guard cell left of column 0), exchange slots neutral
    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = tid; i < m * m1; i += NT) { const int bb = i / m1, qc = i - bb * m1; s_mx[i] = qc < m ? g_mat[bb * m + qc] : 0; } }
    for (int i = tid; i < RR * NPW * RCS; i += NT) {
        const int pl = (i / RCS) % NPW, x = i % RCS - 2;
        const int hz = (int)((unsigned)inf << 16);
        const int infw = (int)(((unsigned)inf & 0xffffu) | ((unsigned)inf << 16));
        fr[i] = (pl == 0 && x == -1) ? hz : (pl == 0 ? infw : inf);
    }
    if (tid < 2 * NW) xch[tid] = make_int4(INT_MIN, INT_MIN, 0, 0);
    __syncthreads();

    T *const planes = io.planes + (long long)lane * CW;
    const int row_stride = nvr * PN * CW;
    // ---- row 0: every plane 0
    if (status == 0) {
#pragma unroll
        for (int c = 0; c < NCW; ++c) if (c < cnt) {
            T *H = planes + (c0 + c) * 64 * CW;
            if (GAP == 1) { *(int2 *)H = make_int2(0, 0); } else { *(int4 *)H = make_int4(0, 0, 0, 0); }
            int *qd0 = fr + 2 + (c0 + c) * 64 + lane;
            qd0[0] = 0; if (NPW > 1) qd0[RCS] = 0;
        }
    }
    __syncthreads();
    // ---- static metadata, two tiles ahead (every wavefront keeps its own copy: the loads are a few hundred bytes per tile)
    struct MetaA { int ps, pe, base; };
    struct MetaB { int p[8]; };
    auto load_a = [&](int t0) __attribute__((always_inline)) { MetaA a; const int r = imin(t0 + lane, gn - 1); a.ps = io.pred_off[r]; a.pe = io.pred_off[r + 1]; a.base = io.row_base[r]; return a; };
    auto load_b = [&](const MetaA &a) __attribute__((always_inline)) {
        MetaB q; const int np = a.pe - a.ps;
#pragma unroll
        for (int k = 0; k < 8; ++k) q.p[k] = io.pred_row[a.ps + imin(k, imax(np - 1, 0))];
        return q;
    };
    MetaA a1 = load_a(0); MetaB b1 = load_b(a1); MetaA a2 = load_a(64);
    int tv_meta = 0, tv_ps = 0, tv_p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int vg_mi = 0;
    int best_score = inf, best_i = 0, best_j = 0, n_rows_done = 0;
    int qoff[NCW];
#pragma unroll
    for (int c = 0; c < NCW; ++c) { const int col = (c0 + c) * 64 + lane; qoff[c] = (col >= 1 && col <= qlen) ? (int)s_query[imin(col - 1, imax(qlen - 1, 0))] : m; }

    for (int t0 = 0; t0 < gn - 1 && status == 0; t0 += 64) {
        if (t0 > 0 && wid == 0) { const int rb = t0 - 64 + lane; io.g_bsn[rb] = 0; io.g_esn[rb] = end_sn; io.g_coff[rb] = (long long)rb * row_stride; io.row_max_i[rb] = vg_mi; }
        {
            const int np_ = a1.pe - a1.ps;
            tv_meta = (a1.base & 0xff) | (imin(np_, 255) << 8); tv_ps = a1.ps;
#pragma unroll
            for (int k = 0; k < 8; ++k) tv_p[k] = b1.p[k];
            a1 = a2; b1 = load_b(a1); a2 = load_a(t0 + 128);
        }
        const int r_hi = imin(t0 + 64, gn - 1);
        for (int row = imax(t0, 1); row < r_hi; ++row) {
            const int ti = row & 63;
            const int meta = __builtin_amdgcn_readlane(tv_meta, ti), base = meta & 0xff, np = (meta >> 8) & 0xff;
            const int *mrow = s_mx + base * m1;
            int q[NCW], Mv[NCW], E1v[NCW], E2v[NCW], kb[NCW];
#pragma unroll
            for (int c = 0; c < NCW; ++c) q[c] = mrow[qoff[c]];
            // ---- predecessor gather for this wavefront's chunks (a chunk it does not have -- c >= cnt -- reads its neighbour's cells and drops them)
            auto read_pred = [&](int p, int *hc, int *ec1, int *ec2) __attribute__((always_inline)) {
                if (row - p < RR) {
                    const int *src = ring_at(__builtin_amdgcn_readlane(vslot, p), lane - 1);
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        const int co = (c < cnt ? c0 + c : 0) * 64;      // (a chunk slot this wavefront does not use reads chunk 0: always inside the ring row)
                        const int w0 = src[co], w1 = src[co + 1]; hc[c] = (int)(short)w0; ec1[c] = w1 >> 16; ec2[c] = GAP == 2 ? src[RCS + co + 1] : inf;
                    }
                } else {
                    // older than the ring: from the arena.  The diagonal cell at the chunk boundary was stored by the neighbour wavefront: every wavefront drains
                    // its stores, then the team meets (the condition is the same in all of them)
                    const T *Hp = io.planes + (long long)p * row_stride;
                    gld_wait();
                    lds_barrier();
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        const int x = (c0 + c) * 64 + lane, xh = med3i(x - 1, 0, W - 1), xe = imin(x, W - 1);
                        gld_async_cell(hc[c], Hp + (long long)xh * CW); gld_async_cell(ec1[c], Hp + (long long)xe * CW + PL_E1);
                        if (GAP == 2) gld_async_cell(ec2[c], Hp + (long long)xe * CW + PL_E2); else ec2[c] = inf;
                    }
#pragma unroll
                    for (int c = 0; c < NCW; ++c) { if (GAP == 2) asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]), "+v"(ec2[c]) :: "memory");
                            else asm volatile("s_waitcnt vmcnt(0)" : "+v"(hc[c]), "+v"(ec1[c]) :: "memory"); }
                    if (lane == 0 && c0 == 0) hc[0] = 0;             // H_p[-1] = 0
                }
            };
            {
                int hc[NCW], ec1[NCW], ec2[NCW];
                read_pred(__builtin_amdgcn_readlane(tv_p[0], ti), hc, ec1, ec2);
#pragma unroll
                for (int c = 0; c < NCW; ++c) { Mv[c] = hc[c]; E1v[c] = ec1[c]; E2v[c] = ec2[c]; kb[c] = 1; }
                auto another = [&](int p, int kidx) __attribute__((always_inline)) {
                    int hd[NCW], ed1[NCW], ed2[NCW];
                    read_pred(p, hd, ed1, ed2);
#pragma unroll
                    for (int c = 0; c < NCW; ++c) { kb[c] = hd[c] > Mv[c] ? kidx : kb[c]; Mv[c] = imax(Mv[c], hd[c]); E1v[c] = imax(E1v[c], ed1[c]); if (GAP == 2) E2v[c] = imax(E2v[c], ed2[c]); }
                };
                if (np > 1) {
                    another(__builtin_amdgcn_readlane(tv_p[1], ti), 2);
                    for (int k = 2; k < np; ++k) {      // (a run-time loop from the third predecessor on: code size)
                        const int tvk = k == 2 ? tv_p[2] : (k == 3 ? tv_p[3] : (k == 4 ? tv_p[4] : (k == 5 ? tv_p[5] : (k == 6 ? tv_p[6] : tv_p[7]))));
                        const int p = k < 8 ? __builtin_amdgcn_readlane(tvk, ti) : __builtin_amdgcn_readfirstlane(gld_i32(io.pred_row + __builtin_amdgcn_readlane(tv_ps, ti) + k));
                        another(p, imin(k + 1, 65));
                    }
                }
            }
            // ---- H before F; per-chunk unseeded prefix maxima; arg-max key from max(0, M + q, E)
            int h[NCW], hs[NCW], hsE[NCW], g1[NCW], g2[NCW], s1[NCW], s2[NCW];
            unsigned amk = 0;
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                h[c] = wr(Mv[c] + q[c]);
                hs[c] = h[c]; if (GAP == 2) hs[c] = imax(imax(h[c], E1v[c]), E2v[c]);
                hsE[c] = GAP == 1 ? imax(h[c], E1v[c]) : hs[c];
                g1[c] = hs[c] + le1; s1[c] = wave_shr1(INT_MIN, g1[c]);
                if (GAP == 2) { g2[c] = hs[c] + le2; s2[c] = wave_shr1(INT_MIN, g2[c]); } else { g2[c] = 0; s2[c] = 0; }
                const int col = (c0 + c) * 64 + lane, vb = (c0 + c) * NV;
                const bool in_band = c < cnt && col < W, is_end = (vb + vvl == end_sn);
                const int cand = (is_end && col > qlen) ? inf : imax(0, hsE[c]);
                const unsigned key = ((unsigned)cand << 16) + (unsigned)(kconst - vb) + (is_end ? 2048u : 0u);
                amk = (in_band && key > amk) ? key : amk;
            }
            {
                auto step = [&](auto ctrl, auto rmask) __attribute__((always_inline)) {
                    constexpr int CT = decltype(ctrl)::value, RM = decltype(rmask)::value;
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
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
            // ---- carry chain over this wavefront's chunk totals as if nothing came in (wavefront 0 starts from the row's first column: first - e)
            int seed1[NCW + 1], seed2[NCW + 1];
            seed1[0] = wid == 0 ? __builtin_amdgcn_readlane(h[0], 0) - e1 : INT_MIN; seed2[0] = wid == 0 ? seed1[0] + e1 - e2 : INT_MIN;
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                seed1[c + 1] = imax(__builtin_amdgcn_readlane(imax(s1[c], g1[c]), 63), seed1[c]) - 64 * e1;
                if (GAP == 2) seed2[c + 1] = imax(__builtin_amdgcn_readlane(imax(s2[c], g2[c]), 63), seed2[c]) - 64 * e2; else seed2[c + 1] = INT_MIN;
            }
            {
                int out1 = seed1[0], out2 = seed2[0];
#pragma unroll
                for (int c = 0; c < NCW; ++c) if (c < cnt) { out1 = seed1[c + 1]; out2 = seed2[c + 1]; }
                int4 *xs = xch + (row & 1) * NW;
                if (lane == 0) xs[wid] = make_int4(out1, out2, cnt > 0 ? (int)kbst : 0, 0);
                lds_barrier();
                const int4 en = xs[lane & (NW - 1)];
                // incoming seed: the entries of the wavefronts before this one, folded in order (saturating: INT_MIN stays "nothing")
                int in1 = INT_MIN, in2 = INT_MIN; unsigned kall = 0;
#pragma unroll
                for (int j = 0; j < NW; ++j) {
                    const int a1_ = __builtin_amdgcn_readlane(en.x, j), a2_ = __builtin_amdgcn_readlane(en.y, j);
                    const unsigned kj = (unsigned)__builtin_amdgcn_readlane(en.z, j);
                    kall = kj > kall ? kj : kall;
                    if (j < wid) {
                        const int cj = (bs_ + (j < rm_ ? 1 : 0)) * 64;
                        in1 = imax(a1_, imax(in1, INT_MIN + cj * e1) - cj * e1); in2 = imax(a2_, imax(in2, INT_MIN + cj * e2) - cj * e2);
                    }
                }
                kbst = kall;
                if (wid > 0) {
#pragma unroll
                    for (int c = 0; c < NCW; ++c) {
                        seed1[c] = imax(seed1[c], imax(in1, INT_MIN + c * 64 * e1) - c * 64 * e1);
                        if (GAP == 2) seed2[c] = imax(seed2[c], imax(in2, INT_MIN + c * 64 * e2) - c * 64 * e2);
                    }
                }
            }
            // ---- F, H, E of every chunk; records to the arena, H / E to the ring
            T *const Hrow = planes + (long long)row * row_stride + (long long)c0 * 64 * CW;
            int *const qd = ring_at(__builtin_amdgcn_readlane(vslot, ti), c0 * 64 + lane);
#pragma unroll
            for (int c = 0; c < NCW; ++c) {
                const int F1 = imax(imax(s1[c], seed1[c]) - cf1, inj1);
                int F2 = inf; if (GAP == 2) F2 = imax(imax(s2[c], seed2[c]) - cf2, inj2);
                int Hout, E1out, E2out = inf;
                if (GAP == 1) {
                    Hout = imax(0, imax(hsE[c], F1));
                    const int en_ = imax(wr(E1v[c] - e1), wr(Hout - oe1));
                    E1out = (Hout == hsE[c]) ? en_ : 0;
                } else {
                    Hout = imax(0, imax(hs[c], imax(F1, F2)));
                    E1out = imax(0, imax(wr(E1v[c] - e1), wr(Hout - oe1)));
                    E2out = imax(0, imax(wr(E2v[c] - e2), wr(Hout - oe2)));
                }
                const int mflag = (Mv[c] + q[c] == Hout && kb[c] <= 64 && Hout != 0) ? kb[c] : 0;
                const int he = (int)(((unsigned)Hout & 0xffffu) | ((unsigned)E1out << 16));
                if (c < cnt) {      // (lanes past the row's end -- last chunk only -- write cells the next row overwrites: same wavefront, program order)
                    T *H = Hrow + c * 64 * CW;
                    if (GAP == 1) { int2 rec; rec.x = he; rec.y = (int)__builtin_amdgcn_perm((unsigned)mflag, (unsigned)F1, 0x05040100u); *(int2 *)H = rec; }
                    else { int4 rec; rec.x = he; rec.y = (int)(((unsigned)E2out & 0xffffu) | ((unsigned)F1 << 16)); rec.z = F2 & 0xffff; rec.w = mflag; *(int4 *)H = rec; }
                    qd[c * 64] = he; if (GAP == 2) qd[RCS + c * 64] = E2out;
                }
            }
            // ---- row maximum -> best cell (every wavefront keeps the same record; wavefront 0 reports it)
            int rowmax, mi = -1;
            rowmax = (int)(kbst >> 16) - 32768; if (rowmax > inf) { mi = (2047 - (int)(kbst & 0x7ff)) * PN + (PN - 1 - (int)((kbst >> 12) & 0xf)); if (mi > qlen) mi = -1; }
            if (rowmax > best_score) { best_score = rowmax; best_i = row; best_j = mi; }
            { const int mi_s = sgpr(mi); asm volatile("s_mov_b32 m0, %2\n\ts_nop 3\n\tv_writelane_b32 %0, %1, m0" : "+v"(vg_mi) : "s"(mi_s), "s"(ti) : "m0"); }
            n_rows_done = row;
            lds_barrier();                                            // the row's ring slot is complete before any wavefront reads it
        }
    }
    if (status == 0 && wid == 0) {
        const int tb = n_rows_done & ~63, rb = tb + lane;
        if (rb <= n_rows_done) { io.g_bsn[rb] = 0; io.g_esn[rb] = end_sn; io.g_coff[rb] = (long long)rb * row_stride; io.row_max_i[rb] = vg_mi; }
    }
    if (tid == 0) {
        GLOBAL_AS AlnOut *o = vgpr_ptr(out_rec);
        o->status = status; o->n_cells = status == 0 ? (long long)imax(0, gn - 2) * W : 0; o->cells_used = status == 0 ? (long long)(gn - 1) * row_stride : 0;
        o->clk_dp = 0; o->n_rows_done = n_rows_done; o->best_score = best_score; o->best_row = best_i; o->best_col = best_j;
        for (int i_ = 0; i_ < 6; ++i_) o->seg[i_] = 0;
    }
}

}  // namespace abpoa_hip
