#pragma once
#include "dp_common.h"
#include "backtrack.h"

namespace abpoa_hip {

// GAP: 0 linear, 1 affine, 2 convex (reference gap_mode)
template <typename T, int GAP>
__device__ __forceinline__ void align_one(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int P = GAP == 0 ? 1 : (GAP == 1 ? 3 : 5);
    constexpr int NPR = GAP == 0 ? 1 : (GAP == 1 ? 2 : 3);        // planes kept in the LDS score ring: H, E1, E2
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, w = d.w;
    const bool local = b.align_mode == ABPOA_HIP_LOCAL_MODE, extend = b.align_mode == ABPOA_HIP_EXTEND_MODE;
    const bool banded = b.wb >= 0;
    const T inf = (T)d.inf_min;
    const T e1 = (T)b.e1, o1 = (T)b.o1, oe1 = (T)(b.o1 + b.e1), e2 = (T)b.e2, o2 = (T)b.o2, oe2 = (T)(b.o2 + b.e2);
    const int dp_sn = (qlen + PN) / PN;
    // fast F path (fast_f_chain): per-lane constants and the no-wrap threshold
    const int idist = inj_dist<PN>(l);
    const int cl1 = (int)oe1 + l * (int)e1, cl2 = (int)oe2 + l * (int)e2;
    const int inj1 = idist >= 0 ? (int)inf - idist * (int)e1 : INT_MIN, inj2 = idist >= 0 ? (int)inf - idist * (int)e2 : INT_MIN;
    const long long lo_ll = (long long)(sizeof(T) == 2 ? INT16_MIN : INT32_MIN) + imax((int)oe1, (int)oe2) + (long long)PN * imax((int)e1, (int)e2);
    const int fast_lo = (int)lo_ll;

    GLOBAL_AS const uint8_t *g_query = vgpr_ptr(b.query + d.query_off);
    GLOBAL_AS const uint8_t *row_base = vgpr_ptr(b.row_base + d.row0);
    GLOBAL_AS const int32_t *row_node_id = vgpr_ptr(b.row_node_id + d.row0);
    GLOBAL_AS const int32_t *row_remain = vgpr_ptr(b.row_remain + d.row0);
    GLOBAL_AS const uint8_t *row_active = vgpr_ptr(b.row_active + d.row0);
    GLOBAL_AS const int32_t *pred_off = vgpr_ptr(b.pred_off + d.poff0), *pred_row = vgpr_ptr(b.pred_row + d.pred0);
    GLOBAL_AS const int32_t *out_off = vgpr_ptr(b.out_off + d.poff0), *out_row = vgpr_ptr(b.out_row + d.out0);
    GLOBAL_AS int32_t *g_left = vgpr_ptr(b.left + d.row0), *g_right = vgpr_ptr(b.right + d.row0);
    GLOBAL_AS int32_t *g_bsn = vgpr_ptr(b.dp_beg_sn + d.row0), *g_esn = vgpr_ptr(b.dp_end_sn + d.row0), *row_max_i = vgpr_ptr(b.row_max_i + d.row0);
    GLOBAL_AS int64_t *g_coff = vgpr_ptr(b.row_cell_off + d.row0);
    T *planes = (T *)(b.planes + d.plane_off);

    // ---- LDS carve-up (engine.h LdsPlan)
    uint8_t *s_query = lds_raw + b.lds.q_off;
    int32_t *s_mat = (int32_t *)(lds_raw + b.lds.mat_off);
    DpLds &S = *(DpLds *)(lds_raw + b.lds.phase_off);
    T *s_ring = (T *)(lds_raw + b.lds.phase_off + b.lds.ring_off);
    const int ring_rows = b.lds.ring_rows, ring_cols = b.lds.ring_cols;
    const bool q_in_lds = qlen <= b.lds.q_cap;

    { GLOBAL_AS const int32_t *g_mat = vgpr_ptr(b.mat); for (int i = lane; i < m * m; i += 64) s_mat[i] = g_mat[i]; }
    if (q_in_lds) for (int i = lane; i < qlen; i += 64) s_query[i] = g_query[i];

    // dp_end as the reference stores it: vector-rounded when banded and for row 0, qlen otherwise
    auto dp_end_of = [&](int row, int end_sn_row) __attribute__((always_inline)) { return (banded || row == 0) ? (end_sn_row + 1) * PN - 1 : qlen; };

    // literal (wrap-exact) F recurrence for the vectors [nfast_, ...) of chunk c: reference :859-875 / :978-997
    auto slow_f_tail = [&](int c, int beg_sn_, int end_sn_, int max_pre_, int nfast_, T hs, T &F1, T &F2, T &first, T &first2) __attribute__((always_inline)) {
#pragma unroll
        for (int vv = 0; vv < NV; ++vv) {
            const int vg = beg_sn_ + c * NV + vv;
            if (vv >= nfast_ && vg <= end_sn_) {
                int set_num = PN;
                if (!local && vg > max_pre_) set_num = (vg == max_pre_ + 1) ? 2 : 1;
                T prev = (T)row_shr<1>((int)first, (int)hs);
                if (PN == 8) prev = (l == 0) ? first : prev;
                T f = wsub<T>(prev, oe1);                        // reference :870 / :990
                f = set_f<T>(f, l, set_num, e1, inf);
                const T hlast = (T)__builtin_amdgcn_readlane((int)hs, vv * PN + PN - 1);
                first = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)f, vv * PN + PN - 1), o1));  // :874 / :996
                if (vvl == vv) F1 = f;
                if (GAP == 2) {
                    T prev2 = (T)row_shr<1>((int)first2, (int)hs);
                    if (PN == 8) prev2 = (l == 0) ? first2 : prev2;
                    T g = wsub<T>(prev2, oe2);                   // reference :991
                    g = set_f<T>(g, l, set_num, e2, inf);
                    first2 = tmax<T>(hlast, wadd<T>((T)__builtin_amdgcn_readlane((int)g, vv * PN + PN - 1), o2));  // :997
                    if (vvl == vv) F2 = g;
                }
            }
        }
    };

    // Linear gaps (reference :762-778): the row's H is max(h, H[col-1] - e) taken vector by vector with SIMD_SET_F on H itself.  For the vectors that use the plain scan
    // (set_num == pn: up to max_pre_end_sn) and while nothing can wrap, max-plus arithmetic distributes over the whole 64-lane chunk:
    //   H[c] = max( max_{c' <= c} (h[c'] + c' e) - c e ,  first - c e ,  inf )      (first enters at lane 0; the clamp at `inf` is the reference's: `first` holds inf in
    // its lanes 1 .. pn - 1 and `max(H, first)` lifts those lanes before the scan, lane 0 gets inf from the scan's shifted-in lane -- NOT the affine F scan's
    // `inf - INJ e`; tests/test_linear_closed_form.py) -- ONE 64-lane prefix-max scan instead of 64 / pn log-step scans with a readlane between them.
    // The vectors beyond every predecessor's band (set_num 1 / 0) keep the literal masked scan.
    const int le1 = lane * (int)e1;
    auto linear_h = [&](int c, int beg_sn_, int end_sn_, int max_pre_, T h, T &first) __attribute__((always_inline)) -> T {
        if (c == 0) first = (T)__builtin_amdgcn_readlane((int)h, 0);
        const int vb = beg_sn_ + c * NV;
        const int nvec = imin(NV, end_sn_ - vb + 1);
        int nfast = local ? nvec : imin(nvec, max_pre_ - vb + 1);
        if (nfast < 0) nfast = 0;
        if (nfast > 0 && ((int)first < fast_lo || __any(vvl < nfast && (int)h < fast_lo))) nfast = 0;
        if (b.dbg & 4) nfast = 0;
        if (nfast > 0) {
            int g = (int)h + le1;
            g = lane == 0 ? imax(g, (int)first) : g;
            const int Hc = imax(wave_scan_max_i32(g) - le1, (int)inf);      // (the clamp: see the comment above)
            if (vvl < nfast) h = (T)Hc;
            first = (T)(__builtin_amdgcn_readlane(Hc, nfast * PN - 1) - (int)e1);
        }
#pragma unroll
        for (int vv = 0; vv < NV; ++vv) {
            if (vv >= nfast && vb + vv <= end_sn_) {
                const int vg = vb + vv;
                int set_num = PN;
                if (!local && vg > max_pre_) set_num = (vg == max_pre_ + 1) ? 1 : 0;
                T hv = tmax<T>(h, l == 0 ? first : inf);
                hv = set_f<T>(hv, l, set_num, e1, inf);
                if (vvl == vv) h = hv;
                first = wsub<T>((T)__builtin_amdgcn_readlane((int)hv, vv * PN + PN - 1), e1);
            }
        }
        return h;
    };

    long long cursor = 0;          // next free arena cell
    long long n_cells = 0;
    int status = 0;
    int rows_done = 0, bt_steps = 0;
    int best_score = d.inf_min, best_i = 0, best_j = 0, best_row_zd = 0;
    int last_done = 0;                                        // last row the loop reached (z-drop may stop early)
    long long clk0 = 0, clk1 = 0;
#ifdef ABPOA_HIP_PROFILE
    long long seg_keep[6] = {0, 0, 0, 0, 0, 0};
#endif
    {
    // ------------------------------------------------------------------ row 0, reference :553-662
    int end_sn0 = 0;
    {
        int dp_end0;
        if (banded) {
            int r = row_remain[0] - row_remain[gn - 1] - 1;
            dp_end0 = imin(qlen, imax(0, qlen - r) + w);          // max_pos_right[begin] == 0
        } else dp_end0 = qlen;
        end_sn0 = dp_end0 / PN;
        const int W0 = (end_sn0 + 1) * PN;
        if ((long long)W0 * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; }
        else {
            const bool ring0 = W0 <= ring_cols;
            if (lane == 0) { g_bsn[0] = 0; g_esn[0] = end_sn0; g_coff[0] = 0; S.b_rec[0] = make_int4(0, end_sn0, 0, ring0 ? 0 : -1); }
            for (int i = lane; i < W0; i += 64) {
                T h, x1 = inf, x2 = inf, f1 = inf, f2 = inf;
                if (local) { h = 0; x1 = 0; x2 = 0; f1 = 0; f2 = 0; }
                else if (GAP == 0) h = (T)(-(int)e1 * i);
                else if (GAP == 1) {
                    T g = (T)(-(int)o1 - (int)e1 * i);
                    h = i == 0 ? (T)0 : g; x1 = i == 0 ? (T)(-(int)oe1) : inf; f1 = i == 0 ? inf : g;
                } else {
                    T g1 = (T)(-(int)o1 - (int)e1 * i), g2 = (T)(-(int)o2 - (int)e2 * i);
                    h = i == 0 ? (T)0 : tmax<T>(g1, g2);
                    x1 = i == 0 ? (T)(-(int)oe1) : inf; x2 = i == 0 ? (T)(-(int)oe2) : inf;
                    f1 = i == 0 ? inf : g1; f2 = i == 0 ? inf : g2;
                }
                planes[i] = h;
                if (GAP != 0) { planes[(long long)PL_E1 * W0 + i] = x1; planes[(long long)PL_F1 * W0 + i] = f1; }
                if (GAP == 2) { planes[(long long)PL_E2 * W0 + i] = x2; planes[(long long)PL_F2 * W0 + i] = f2; }
                if (ring0) {
                    s_ring[i] = h;
                    if (GAP != 0) s_ring[ring_cols + i] = x1;
                    if (GAP == 2) s_ring[2 * ring_cols + i] = x2;
                }
            }
            cursor = (long long)W0 * P;
        }
    }
    // ---- max_pos_left/right look-ahead window: LDS holds rows [lr_blk, lr_blk + RL)
    int lr_blk = 0;
    // (far_seen: some row beyond the window had its band state written to the HBM copy -- from then on a row entering the window is loaded, not assumed untouched)
    bool far_seen = false;
    if (banded && status == 0) {
        if (b.fresh_band) {       // reference abpoa_topological_sort resets them before every alignment (abpoa_graph.c:303-308)
            for (int i = lane; i < gn; i += 64) { g_left[i] = gn; g_right[i] = 0; }
            for (int i = lane; i < RL; i += 64) S.l_lr[i] = make_int2(gn, 0);
        } else
            for (int i = lane; i < RL; i += 64) { const int r = i; if (r < gn) S.l_lr[i] = make_int2(g_left[r], g_right[r]); }
        __syncthreads();
        if (lane == 0) S.l_lr[0] = make_int2(0, 0);                            // reference :556
        bool far0 = false;
        for (int t = out_off[0] + lane; t < out_off[1]; t += 64) {            // reference :557-561
            const int o = out_row[t];
            if (o >= 0 && row_active[o]) {
                if (o < RL) S.l_lr[o] = make_int2(1, 1); else { g_left[o] = 1; g_right[o] = 1; far0 = true; }
            }
        }
        // (a successor of the source beyond the window -- a read that starts in the middle of the graph: without this the row, when it entered the window, was
        //  taken for untouched and lost its (1, 1); found by tools/fuzz_device_vs_oracle.py on ragged reads in extension mode)
        if (__any(far0)) { far_seen = true; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
    }
    __syncthreads();

#ifdef ABPOA_HIP_PROFILE
    long long seg[6] = {0, 0, 0, 0, 0, 0}, seg_last = 0;
#define STAMP(I) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); long long t_ = (long long)__builtin_amdgcn_s_memtime(); seg[I] += t_ - seg_last; seg_last = t_; }
#else
#define STAMP(I)
#endif
    clk0 = (long long)__builtin_amdgcn_s_memtime();
#ifdef ABPOA_HIP_PROFILE
    seg_last = clk0;
#endif
    const int remain_end = (banded || b.zdrop > 0) ? row_remain[gn - 1] : 0;
    const bool need_max = local || extend || banded;
    int tile_beg = 0, tile_end = 0, pbase = 0, obase = 0;     // static-metadata tile covers rows [tile_beg, tile_end)
    int last_row = 0;                                         // last row whose left/right entry was consumed

    // ------------------------------------------------------------------ rows 1 .. gn-2, reference :1105
    // next-tile prefetch registers (static graph metadata of rows [nt_t0, nt_t0 + TS))
    int4 nt_rec0 = make_int4(0, 0, 0, 0), nt_rec1 = make_int4(0, 0, 0, 0); int nt_pred[TP / 64], nt_out[TP / 64];
    int nt_t0 = 1, nt_pb0 = 0, nt_ob0 = 0;
    // per-lane copy of the CURRENT tile's metadata (lane i <-> row tile_beg + i): the row loop fetches a field with one
    // v_readlane instead of an LDS round trip.  tv_meta = base | active<<8 | fast<<9 | np<<16 | n_out<<24
    int tv_meta = 0, tv_rterm = 0, tv_ps = 0, tv_os = 0, tv_pid[4] = {0, 0, 0, 0}, tv_o[2] = {-1, -1};
    if (gn > 2) {
        nt_pb0 = gld_i32(pred_off + 1); nt_ob0 = gld_i32(out_off + 1);
        const int tend = imin(nt_t0 + TS, gn);
        const int rr = imin(nt_t0 + lane, gn), rc = imin(rr, gn - 1), r2 = imin(nt_t0 + TS, gn);
        nt_rec0.x = pred_off[rr]; nt_rec0.y = out_off[rr];
        const int rem_ = (banded || b.zdrop > 0) ? row_remain[rc] : 0; const int ba_ = (int)row_base[rc] | ((int)row_active[rc] << 8);
        nt_rec0.z = rr < tend ? rem_ : 0; nt_rec0.w = rr < tend ? ba_ : 0;
        nt_rec1.x = pred_off[r2]; nt_rec1.y = out_off[r2];
#pragma unroll
        for (int j = 0; j < TP / 64; ++j) { nt_pred[j] = pred_row[nt_pb0 + j * 64 + lane]; nt_out[j] = out_row[nt_ob0 + j * 64 + lane]; }
    }
    int qc_beg_sn = -1; int qc_cache[2] = {0, 0};            // query codes of this lane's columns for chunks 0/1 of band start qc_beg_sn
    for (int row = 1; row < gn - 1 && status == 0; ++row) {
        if (row >= tile_end) {                                // ---- switch to the next 64-row metadata tile (prefetched in registers)
            // band geometry of the rows of the finished tile goes to HBM in one coalesced burst (backtrack + trace read it)
            if (tile_end > tile_beg && tile_beg + lane < tile_end) {
                const int r = tile_beg + lane; const int4 br = S.b_rec[r % RB];
                g_bsn[r] = br.x; g_esn[r] = br.y; g_coff[r] = (long long)(uint32_t)br.z * PN;
            }
            S.t_rec[lane] = nt_rec0; if (lane == 0) S.t_rec[TS] = nt_rec1;
#pragma unroll
            for (int j = 0; j < TP / 64; ++j) { S.t_pred[j * 64 + lane] = nt_pred[j]; S.t_out[j * 64 + lane] = nt_out[j]; }
            tile_beg = nt_t0; tile_end = imin(nt_t0 + TS, gn); pbase = nt_pb0; obase = nt_ob0;
            {
                const int my_ps = nt_rec0.x, my_os = nt_rec0.y;
                const int4 nx = S.t_rec[lane + 1];
                const int np_ = nx.x - my_ps, on_ = nx.y - my_os;
                bool ok = np_ >= 1 && np_ <= 4 && on_ >= 0 && on_ <= 2;
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int idx = my_ps - pbase + imin(kk, imax(np_ - 1, 0));
                    ok = ok && idx >= 0 && idx < TP; tv_pid[kk] = S.t_pred[(idx >= 0 && idx < TP) ? idx : 0];
                }
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int idx = my_os - obase + kk; const bool has = kk < on_;
                    ok = ok && (!has || (idx >= 0 && idx < TP)); tv_o[kk] = has ? S.t_out[(idx >= 0 && idx < TP) ? idx : 0] : -1;
                }
                tv_meta = (nt_rec0.w & 0x1ff) | (ok ? (1 << 9) : 0) | (imin(imax(np_, 0), 255) << 16) | (imin(imax(on_, 0), 255) << 24);
                tv_rterm = qlen - (nt_rec0.z - remain_end - 1); tv_ps = my_ps; tv_os = my_os;
                // fast-row record (static eligibility: active, 1-2 predecessors within score-ring distance, 1-2 successors)
                const int myrow = tile_beg + lane;
                const int d0 = myrow - tv_pid[0], d1 = myrow - tv_pid[np_ >= 2 ? 1 : 0];
                const bool fok = ok && ((nt_rec0.w >> 8) & 1) && np_ <= 2 && on_ >= 1 && d0 >= 1 && d1 >= 1 && d0 < ring_rows && d1 < ring_rows && d0 < 256 && d1 < 256;
                S.t_fast[lane] = make_int4((fok ? (int)0x80000000u : 0) | ((nt_rec0.w & 0xff) << 16) | ((d1 & 0xff) << 8) | (d0 & 0xff), tv_rterm, tv_o[0], tv_o[1]);
            }
            // issue the loads of the tile after this one right away; they complete while this tile is being processed
            nt_t0 = tile_end; nt_pb0 = __builtin_amdgcn_readfirstlane(nt_rec1.x); nt_ob0 = __builtin_amdgcn_readfirstlane(nt_rec1.y);
            if (nt_t0 < gn - 1) {
                const int tend = imin(nt_t0 + TS, gn);
                const int rr = imin(nt_t0 + lane, gn), rc = imin(rr, gn - 1), r2 = imin(nt_t0 + TS, gn);
                nt_rec0.x = pred_off[rr]; nt_rec0.y = out_off[rr];
                const int rem_ = (banded || b.zdrop > 0) ? row_remain[rc] : 0; const int ba_ = (int)row_base[rc] | ((int)row_active[rc] << 8);
                nt_rec0.z = rr < tend ? rem_ : 0; nt_rec0.w = rr < tend ? ba_ : 0;
                nt_rec1.x = pred_off[r2]; nt_rec1.y = out_off[r2]; nt_rec1.z = 0; nt_rec1.w = 0;
#pragma unroll
                for (int j = 0; j < TP / 64; ++j) { nt_pred[j] = pred_row[nt_pb0 + j * 64 + lane]; nt_out[j] = out_row[nt_ob0 + j * 64 + lane]; }
            }
        }
        if (banded && row >= lr_blk + RLH) {                  // ---- slide the left/right window by half
            // rows [lr_blk, lr_blk+RLH) are retired: write them back, then bring in rows [lr_blk+RL, lr_blk+RL+RLH)
            for (int i = lane; i < RLH; i += 64) {
                const int r = lr_blk + i;
                if (r < gn) { const int2 v2 = S.l_lr[r % RL]; g_left[r] = v2.x; g_right[r] = v2.y; }
                const int nr = lr_blk + RL + i;
                if (nr < gn) {
                    if (b.fresh_band && !far_seen) S.l_lr[nr % RL] = make_int2(gn, 0);       // untouched so far: known without a load
                    else S.l_lr[nr % RL] = make_int2(gld_i32(g_left + nr), gld_i32(g_right + nr));
                }
            }
            lr_blk += RLH;
        }
        STAMP(5)
        last_done = row;
        const int ti = row - tile_beg;
        // ====================================================================================================
        // FAST ROW (the common case of a POA graph): global + banded + affine/convex, one or two predecessors whose H/E
        // rows are still in the score ring, band <= 128 columns, successors inside the left/right window.  Straight-line:
        // one record read, two geometry reads, one batch of score reads per chunk, one reduction.  Same arithmetic as the
        // general row below (which handles everything else), so the results are identical.
        if (GAP != 0 && banded && !local && !extend && !(b.dbg & 64)) {
            const int4 fr = uniform4(S.t_fast[ti]);
            if (fr.x < 0) {
                const int2 lr = uniform2(S.l_lr[row % RL]);
                const int fp0 = row - (fr.x & 0xff), fp1 = row - ((fr.x >> 8) & 0xff), fbase = (fr.x >> 16) & 0xff;
                const int4 g0 = uniform4(S.b_rec[fp0 % RB]), g1 = uniform4(S.b_rec[fp1 % RB]);
                const int so0 = fr.z, so1 = fr.w;      // successor rows (-1 = none)
                const int fbeg = imax(0, imin(lr.x, fr.y) - w), fend = imin(qlen, imax(lr.y, fr.y) + w);      // reference :711
                const int fmin_pre = imin(g0.x, g1.x), fmax_pre = imax(g0.y, g1.y);
                const int fbeg_sn = imax(fbeg / PN, fmin_pre), fend_sn = fend / PN;
                const int fWr = (fend_sn - fbeg_sn + 1) * PN;
                const bool feasible = g0.w == fp0 && g1.w == fp1 && fWr <= 128 && fWr <= ring_cols && so0 < lr_blk + RL && so1 < lr_blk + RL &&
                                      q_in_lds && cursor + (long long)fWr * P <= d.plane_cap;
                if (feasible) {
                    const long long off = cursor; cursor += (long long)fWr * P; n_cells += fWr; ++rows_done; last_row = row;
                    if (lane == 0) S.b_rec[row % RB] = make_int4(fbeg_sn, fend_sn, (int)(uint32_t)(off / PN), -1);
                    T *H = planes + off;
                    T *my_ring = s_ring + (long long)(row % ring_rows) * NPR * ring_cols;
                    const T *rp0 = s_ring + (long long)(fp0 % ring_rows) * NPR * ring_cols, *rp1 = s_ring + (long long)(fp1 % ring_rows) * NPR * ring_cols;
                    if (fbeg_sn != qc_beg_sn) {
                        qc_beg_sn = fbeg_sn;
#pragma unroll
                        for (int c2 = 0; c2 < 2; ++c2) { const int cc = fbeg_sn * PN + c2 * 64 + lane; qc_cache[c2] = (cc >= 1 && cc <= qlen) ? (int)s_query[cc - 1] : -1; }
                    }
                    const int pb0 = g0.x * PN, pse0 = (g0.y + 1) * PN - 1, pb1 = g1.x * PN, pse1 = (g1.y + 1) * PN - 1;   // stored column ranges
                    const int bs0 = imax(g0.x, fbeg_sn), bs1 = imax(g1.x, fbeg_sn);
                    const int esh0 = imin(imin(g0.y + 1, fend_sn), dp_sn - 1), esh1 = imin(imin(g1.y + 1, fend_sn), dp_sn - 1);
                    const int ese0 = imin(g0.y, fend_sn), ese1 = imin(g1.y, fend_sn);
                    T first = 0, first2 = 0; int dbgv = 0;
                    int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false; unsigned am_key = 0;
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        if (c == 1 && fWr <= 64) break;
                        const int rel = c * 64 + lane;
                        const bool in_band = rel < fWr;
                        const int col = fbeg_sn * PN + rel, v = fbeg_sn + c * NV + vvl;
                        const int qc = c == 0 ? qc_cache[0] : qc_cache[1];
                        const int qv = s_mat[fbase * m + (qc >= 0 ? qc : 0)];
                        const int x0 = col - 1 - pb0, x1 = col - 1 - pb1;
                        const bool ok0 = x0 >= 0 && col - 1 <= pse0, ok1 = x1 >= 0 && col - 1 <= pse1;
                        const bool inH0 = in_band && v >= bs0 && v <= esh0, inH1 = in_band && v >= bs1 && v <= esh1;
                        const bool inE0 = in_band && v >= bs0 && v <= ese0, inE1 = in_band && v >= bs1 && v <= ese1;
                        const T h0 = rp0[ok0 ? x0 : 0], h1 = rp1[ok1 ? x1 : 0];
                        const T q = (in_band && qc >= 0) ? (T)qv : (T)0;
                        const T a0 = rp0[ring_cols + (inE0 ? x0 + 1 : 0)], a1 = rp1[ring_cols + (inE1 ? x1 + 1 : 0)];
                        T c0 = 0, c1 = 0;
                        if (GAP == 2) { c0 = rp0[2 * ring_cols + (inE0 ? x0 + 1 : 0)]; c1 = rp1[2 * ring_cols + (inE1 ? x1 + 1 : 0)]; }
                        T Mv = inH0 ? (ok0 ? h0 : inf) : inf;
                        Mv = inH1 ? tmax<T>(Mv, ok1 ? h1 : inf) : Mv;
                        T E1v = inE0 ? a0 : inf; E1v = inE1 ? tmax<T>(E1v, a1) : E1v;
                        T E2v = inf; if (GAP == 2) { E2v = inE0 ? c0 : inf; E2v = inE1 ? tmax<T>(E2v, c1) : E2v; }
                        const T h = wadd<T>(Mv, q);
                        T hs = h; if (GAP == 2) hs = tmax<T>(tmax<T>(h, E1v), E2v);
                        if (c == 0) { first = (T)__builtin_amdgcn_readlane((int)h, 0); first2 = first; }
                        const int nvec = imin(NV, fend_sn - (fbeg_sn + c * NV) + 1);
                        int nfast = imin(nvec, fmax_pre - (fbeg_sn + c * NV) + 1);
                        if (nfast < 0) nfast = 0;
                        if (nfast > 0 && __any(vvl < nfast && (int)h < fast_lo)) nfast = 0;
                        if (b.dbg & 4) nfast = 0;
                        T F1 = inf, F2 = inf;
                        if (nfast > 0) {
                            int fi = (int)first;
                            int dcv = 0;
                            F1 = (T)fast_f_chain<T>((int)hs, fi, nfast, l, vvl, (int)oe1, (int)e1, (int)o1, cl1, inj1, &dcv);
                            if (b.dbg & 512) F1 = (T)dcv;
                            first = (T)fi;
                            if (GAP == 2) { int fi2 = (int)first2; F2 = (T)fast_f_chain<T>((int)hs, fi2, nfast, l, vvl, (int)oe2, (int)e2, (int)o2, cl2, inj2); first2 = (T)fi2; }
                        }
                        if ((b.dbg & 256) && c == 0) dbgv = ((int)first & 0xffff) | (nfast << 16) | (nvec << 20) | ((__builtin_amdgcn_readlane((int)hs, 15) & 0xff) << 24);
                        if (nfast < nvec) slow_f_tail(c, fbeg_sn, fend_sn, fmax_pre, nfast, hs, F1, F2, first, first2);
                        T Hout, E1out, E2out = 0;
                        if (GAP == 1) {                                          // reference :876-883
                            const T tmp = tmax<T>(h, E1v);
                            Hout = tmax<T>(tmp, F1);
                            const T en = tmax<T>(wsub<T>(E1v, e1), wsub<T>(Hout, oe1));
                            E1out = (Hout == tmp) ? en : inf;
                        } else {                                                 // reference :1004-1007
                            Hout = tmax<T>(hs, tmax<T>(F1, F2));
                            E1out = tmax<T>(wsub<T>(E1v, e1), wsub<T>(Hout, oe1));
                            E2out = tmax<T>(wsub<T>(E2v, e2), wsub<T>(Hout, oe2));
                        }
                        if (in_band) {
                            H[rel] = Hout; H[PL_E1 * fWr + rel] = E1out; H[PL_F1 * fWr + rel] = F1;
                            if (GAP == 2) { H[PL_E2 * fWr + rel] = E2out; H[PL_F2 * fWr + rel] = F2; }
                            my_ring[rel] = Hout; my_ring[ring_cols + rel] = E1out;
                            if (GAP == 2) my_ring[2 * ring_cols + rel] = E2out;
                            const bool is_end = (v == fend_sn);
                            int cand = (int)Hout;
                            if (is_end && fend_sn == qlen / PN && col > qlen) cand = (int)inf;
                            if (sizeof(T) == 2) {
                                const unsigned key = ((unsigned)(cand + 32768) << 16) | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)is_end << 11) | (unsigned)(2047 - v);
                                am_key = key > am_key ? key : am_key;
                            } else if (!am_any || (is_end ? cand >= am_val : cand > am_val)) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
                        }
                    }
                    if (lane == 0) S.b_rec[row % RB].w = row;
                    int mi = -1;
                    if (sizeof(T) == 2) {
                        const unsigned kb = (b.dbg & 128) ? wave_max_u32(am_key) : wave_max_u32_b(am_key);
                        const int vmax = (int)(kb >> 16) - 32768;
                        if (vmax > d.inf_min) { mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
                    } else {
                        const int vmax = wave_max_i32(am_any ? am_val : INT_MIN);
                        if (vmax > d.inf_min) {
                            unsigned key = 0;
                            if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                            const unsigned kb = wave_max_u32_b(key);
                            mi = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu)) * PN + (PN - 1 - (int)(kb >> 27));
                            if (mi > qlen) mi = -1;
                        }
                    }
                    if (b.want_trace && lane == 0) row_max_i[row] = (b.dbg & 256) ? dbgv : mi;
                    {                                                            // reference :1059-1067
                        const int out_i = mi + 1;
                        int2 a2 = S.l_lr[(so0 >= 0 ? so0 : 0) % RL], b2 = S.l_lr[(so1 >= 0 ? so1 : 0) % RL];
                        a2.x = imin(a2.x, out_i); a2.y = imax(a2.y, out_i); b2.x = imin(b2.x, out_i); b2.y = imax(b2.y, out_i);
                        if (so0 >= 0 && lane == 0) S.l_lr[so0 % RL] = a2;
                        if (so1 >= 0 && lane == 0) S.l_lr[so1 % RL] = b2;
                    }
                    continue;
                }
            }
        }
        // ==================================================================================================== general row
        const int meta = __builtin_amdgcn_readlane(tv_meta, ti);
        if (!((meta >> 8) & 1)) { if (lane == 0) S.b_rec[row % RB] = make_int4(-1, -1, (int)(uint32_t)(cursor / PN), -1); continue; }
        const int2 lr = S.l_lr[row % RL];
        const int base = meta & 0xff;
        const bool fastmeta = (meta >> 9) & 1;
        const int rterm = __builtin_amdgcn_readlane(tv_rterm, ti);       // qlen - (remain[row] - remain[end] - 1), reference abpoa_align.h:34-35
        const int ps = __builtin_amdgcn_readlane(tv_ps, ti), os = __builtin_amdgcn_readlane(tv_os, ti);
        auto pred_at = [&](int idx) __attribute__((always_inline)) { const int t = idx - pbase; int v = S.t_pred[t < TP ? t : 0]; if (t >= TP) v = gld_i32(pred_row + idx); return v; };
        // band geometry of an earlier row: LDS ring for the last RB rows, HBM copy otherwise (w = -1: never in the score ring)
        auto geom4 = [&](int p) __attribute__((always_inline)) {
            int4 g4 = S.b_rec[p % RB];
            if (row - p >= RB) { g4.x = gld_i32(g_bsn + p); g4.y = gld_i32(g_esn + p); g4.z = (int)(uint32_t)(gld_i64(g_coff + p) / PN); g4.w = -1; }
            return g4;
        };
        // the first (up to) four predecessors are handled in one batch; np <= 4 covers practically every POA node
        int np, on, pid[4]; int4 pg[4];
        if (fastmeta) {
            np = (meta >> 16) & 0xff; on = (meta >> 24) & 0xff;
#pragma unroll
            for (int k = 0; k < 4; ++k) pid[k] = __builtin_amdgcn_readlane(tv_pid[k], ti);
        } else {
            np = S.t_rec[ti + 1].x - ps; on = S.t_rec[ti + 1].y - os;
#pragma unroll
            for (int k = 0; k < 4; ++k) pid[k] = pred_at(ps + imin(k, np - 1));
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) pg[k] = geom4(pid[k]);
        int beg_sn, end_sn, max_pre_end_sn;
        if (!banded) { beg_sn = 0; end_sn = qlen / PN; max_pre_end_sn = end_sn; }        // reference :706-709
        else {                                                                          // reference :710-720
            last_row = row;
            int beg = imax(0, imin(lr.x, rterm) - w), end = imin(qlen, imax(lr.y, rterm) + w);
            beg_sn = beg / PN;
            int min_pre_beg_sn = imin(imin(pg[0].x, pg[1].x), imin(pg[2].x, pg[3].x));          // duplicates of the last one are harmless
            max_pre_end_sn = imax(imax(pg[0].y, pg[1].y), imax(pg[2].y, pg[3].y));
            for (int k = 4; k < np; ++k) { const int4 g4 = geom4(pred_at(ps + k)); min_pre_beg_sn = imin(min_pre_beg_sn, g4.x); max_pre_end_sn = imax(max_pre_end_sn, g4.y); }
            if (beg_sn < min_pre_beg_sn) beg_sn = min_pre_beg_sn;
            end_sn = end / PN;
        }
        const int Wr = (end_sn - beg_sn + 1) * PN;
        const long long off = cursor;
        if (off + (long long)Wr * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
        cursor += (long long)Wr * P;
        n_cells += Wr; ++rows_done;
        const bool to_ring = Wr <= ring_cols;
        if (lane == 0) S.b_rec[row % RB] = make_int4(beg_sn, end_sn, (int)(uint32_t)(off / PN), -1);   // score-ring tag set when the row is complete
        T *H = planes + off;
        const int my_slot = row % ring_rows;
        T *my_ring = s_ring + (long long)my_slot * NPR * ring_cols;
        const int nchunk = (b.dbg & 8) ? 0 : (Wr + 63) >> 6;
        T first = 0, first2 = 0;
        // running arg-max state of this lane (reference :1043-1057)
        int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false;
        unsigned am_key = 0;          // int16: value and tie-break priority packed into one word (value<<16 | 15-lane<<12 | is_end<<11 | 2047-vector)
        // fast gather: every predecessor's H/E row is in the LDS score ring -> straight-line, batched LDS reads
        bool all_ring = np <= 4;
#pragma unroll
        for (int k = 0; k < 4; ++k) all_ring = all_ring && (row - pid[k] < ring_rows) && (pg[k].w == pid[k]);
        if (q_in_lds && beg_sn != qc_beg_sn) {                // band start moved: refresh this lane's cached query codes
            qc_beg_sn = beg_sn;
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) { const int cc = beg_sn * PN + c2 * 64 + lane; qc_cache[c2] = (cc >= 1 && cc <= qlen) ? (int)s_query[cc - 1] : -1; }
        }

        STAMP(0)
        for (int c = 0; c < nchunk; ++c) {
            const int rel = c * 64 + lane;
            const bool in_band = rel < Wr;
            const int col = beg_sn * PN + rel;
            const int v = beg_sn + c * NV + vvl;
            T Mv = inf, E1v = inf, E2v = inf;
            // query profile value, reference :504-510
            T q = 0;
            if (!(b.dbg & 32)) {
                int qc;
                if (q_in_lds) qc = c == 0 ? qc_cache[0] : c == 1 ? qc_cache[1] : ((col >= 1 && col <= qlen) ? (int)s_query[col - 1] : -1);
                else qc = (in_band && col >= 1 && col <= qlen) ? gld_u8(g_query + col - 1) : -1;
                const int qv = s_mat[base * m + (qc >= 0 ? qc : 0)];
                q = (in_band && qc >= 0) ? (T)qv : (T)0;
            }
            // ---- predecessors, reference :722-761 / :803-852 / :912-969
            if (b.dbg & 16) { Mv = (T)(col & 15); E1v = inf; }
            else if (all_ring) {
                auto gather_ring = [&](auto npc) __attribute__((always_inline)) {
                    constexpr int N = decltype(npc)::value;
                    int hraw[N], e1raw[N], e2raw[N], vraw[N]; bool inHk[N], inEk[N], srcok[N], vok[N];
#pragma unroll
                    for (int k = 0; k < N; ++k) {                      // issue every LDS read first (clamped addresses), select afterwards
                        const int pb = pg[k].x, pe = pg[k].y;
                        const int p_stored_end = (pe + 1) * PN - 1;
                        int bs, es_h, es_e;
                        if (local) { bs = 0; es_h = end_sn; es_e = end_sn; }
                        else {
                            bs = pb < beg_sn ? beg_sn : pb;
                            es_h = imin(imin((dp_end_of(pid[k], pe) + 1) / PN, end_sn), dp_sn - 1);
                            es_e = imin(pe, end_sn);
                        }
                        inHk[k] = in_band && v >= bs && v <= es_h;
                        inEk[k] = GAP != 0 && in_band && v >= bs && v <= es_e;
                        const T *rp = s_ring + (long long)(pid[k] % ring_rows) * NPR * ring_cols;
                        const int x = col - 1 - pb * PN;                // column col-1 relative to the source row's band start
                        srcok[k] = x >= 0 && col - 1 <= p_stored_end;
                        hraw[k] = (int)rp[srcok[k] ? x : 0];
                        if (GAP == 0) { vok[k] = col <= p_stored_end; vraw[k] = (int)rp[vok[k] ? x + 1 : 0]; }
                        if (GAP != 0) e1raw[k] = (int)rp[ring_cols + (inEk[k] ? x + 1 : 0)];
                        if (GAP == 2) e2raw[k] = (int)rp[2 * ring_cols + (inEk[k] ? x + 1 : 0)];
                    }
#pragma unroll
                    for (int k = 0; k < N; ++k) {
                        T hval = srcok[k] ? (T)hraw[k] : (local ? (T)0 : inf);
                        if (GAP == 0) { const T vert = vok[k] ? (T)vraw[k] : inf; hval = tmax<T>(wadd<T>(hval, q), wsub<T>(vert, e1)); }
                        if (k == 0) Mv = inHk[k] ? hval : inf; else Mv = inHk[k] ? tmax<T>(Mv, hval) : Mv;
                        if (GAP != 0) { if (k == 0) E1v = inEk[k] ? (T)e1raw[k] : inf; else E1v = inEk[k] ? tmax<T>(E1v, (T)e1raw[k]) : E1v; }
                        if (GAP == 2) { if (k == 0) E2v = inEk[k] ? (T)e2raw[k] : inf; else E2v = inEk[k] ? tmax<T>(E2v, (T)e2raw[k]) : E2v; }
                    }
                };
                if (np == 1) gather_ring(std::integral_constant<int, 1>{});
                else if (np == 2) gather_ring(std::integral_constant<int, 2>{});
                else if (np == 3) gather_ring(std::integral_constant<int, 3>{});
                else gather_ring(std::integral_constant<int, 4>{});
            } else
            for (int k = 0; k < np; ++k) {
                const int p = pred_at(ps + k);
                const int4 g4 = geom4(p);
                const int pb = g4.x, pe = g4.y; const long long poff = (long long)(uint32_t)g4.z * PN;
                const int Wp = (pe - pb + 1) * PN;
                const int p_stored_end = (pe + 1) * PN - 1;          // last stored column of the predecessor row
                const int pslot = p % ring_rows;
                const bool in_ring = (row - p < ring_rows) && g4.w == p;
                int bs, es_h, es_e;
                if (local) { bs = 0; es_h = end_sn; es_e = end_sn; }
                else {
                    bs = pb < beg_sn ? beg_sn : pb;
                    es_h = imin(imin((dp_end_of(p, pe) + 1) / PN, end_sn), dp_sn - 1);
                    es_e = imin(pe, end_sn);
                }
                const bool inH = in_band && v >= bs && v <= es_h;
                const bool inE = GAP != 0 && in_band && v >= bs && v <= es_e;
                // The source row is read either from the LDS score ring or from its HBM copy; the two paths are
                // instantiated separately so that no generic (flat) pointer is ever formed.
                auto gather = [&](auto from_lds, const T *Hp, const long long pstride) __attribute__((always_inline)) {
                    auto ld = [&](const T *p_) __attribute__((always_inline)) -> T { if constexpr (decltype(from_lds)::value) return *p_; else return (T)gld_cell((GLOBAL_AS const T *)p_); };
                    if (inH) {
                        T hval;
                        if (col == bs * PN) {
                            if (local) hval = 0;
                            else hval = (pb < beg_sn && beg_sn * PN - 1 <= p_stored_end) ? ld(Hp + beg_sn * PN - 1 - pb * PN) : inf;
                        } else hval = (col - 1 <= p_stored_end) ? ld(Hp + col - 1 - pb * PN) : inf;
                        if (GAP == 0) {
                            T vert = (col <= p_stored_end) ? ld(Hp + col - pb * PN) : inf;
                            hval = tmax<T>(wadd<T>(hval, q), wsub<T>(vert, e1));
                        }
                        Mv = (k == 0) ? hval : tmax<T>(Mv, hval);
                    }
                    if (inE) {
                        T ev = ld(Hp + (long long)PL_E1 * pstride + col - pb * PN);
                        E1v = (k == 0) ? ev : tmax<T>(E1v, ev);
                        if (GAP == 2) {
                            T ev2 = ld(Hp + (long long)PL_E2 * pstride + col - pb * PN);
                            E2v = (k == 0) ? ev2 : tmax<T>(E2v, ev2);
                        }
                    }
                };
                if (in_ring) gather(std::true_type{}, s_ring + (long long)pslot * NPR * ring_cols, (long long)ring_cols);
                else gather(std::false_type{}, planes + poff, (long long)Wp);       // HBM copy (older or over-wide row)
            }
            STAMP(1)
            // ---- in-row part
            T Hout, E1out = 0, E2out = 0, F1 = inf, F2 = inf;
            if (GAP == 0) {
                // reference :762-778
                const T h = linear_h(c, beg_sn, end_sn, max_pre_end_sn, Mv, first);
                Hout = local ? tmax<T>((T)0, h) : h;
            } else {
                T h = wadd<T>(Mv, q);                                   // reference :854-856 / :972-974
                T hs = h;                                               // value the F recurrence opens from
                if (GAP == 2) hs = tmax<T>(tmax<T>(h, E1v), E2v);       // reference :988
                if (c == 0) { first = (T)__builtin_amdgcn_readlane((int)h, 0); first2 = first; }   // :858 / :976-977
                // leading vectors that use the plain scan (set_num == pn) go through the closed form when nothing can wrap
                const int nvec = imin(NV, end_sn - (beg_sn + c * NV) + 1);
                int nfast = local ? nvec : imin(nvec, max_pre_end_sn - (beg_sn + c * NV) + 1);
                if (nfast < 0) nfast = 0;
                if (nfast > 0 && __any(vvl < nfast && (int)h < fast_lo)) nfast = 0;
                if (b.dbg & 4) nfast = 0;
                if (nfast > 0) {
                    int fi = (int)first;
                    F1 = (T)fast_f_chain<T>((int)hs, fi, nfast, l, vvl, (int)oe1, (int)e1, (int)o1, cl1, inj1);
                    first = (T)fi;
                    if (GAP == 2) {
                        int fi2 = (int)first2;
                        F2 = (T)fast_f_chain<T>((int)hs, fi2, nfast, l, vvl, (int)oe2, (int)e2, (int)o2, cl2, inj2);
                        first2 = (T)fi2;
                    }
                }
                if (nfast < nvec) slow_f_tail(c, beg_sn, end_sn, max_pre_end_sn, nfast, hs, F1, F2, first, first2);
                if (GAP == 1) {                                          // reference :876-883
                    T tmp = tmax<T>(h, E1v);
                    T hh = tmax<T>(tmp, F1);
                    if (local) hh = tmax<T>((T)0, hh);
                    T en = tmax<T>(wsub<T>(E1v, e1), wsub<T>(hh, oe1));
                    E1out = (hh == tmp) ? en : (local ? (T)0 : inf);
                    Hout = hh;
                } else {                                                 // reference :998-1008
                    T hh = tmax<T>(hs, tmax<T>(F1, F2));
                    if (local) hh = tmax<T>((T)0, hh);
                    E1out = tmax<T>(wsub<T>(E1v, e1), wsub<T>(hh, oe1));
                    E2out = tmax<T>(wsub<T>(E2v, e2), wsub<T>(hh, oe2));
                    if (local) { E1out = tmax<T>((T)0, E1out); E2out = tmax<T>((T)0, E2out); }
                    Hout = hh;
                }
            }
            STAMP(2)
            if (in_band && !(b.dbg & 1)) {
                H[rel] = Hout;
                if (GAP != 0) {
                    H[(long long)PL_E1 * Wr + rel] = E1out;
                    H[(long long)PL_F1 * Wr + rel] = F1;
                    if (GAP == 2) { H[(long long)PL_E2 * Wr + rel] = E2out; H[(long long)PL_F2 * Wr + rel] = F2; }
                }
            }
            if (in_band) {
                if (to_ring) {
                    my_ring[rel] = Hout;
                    if (GAP != 0) my_ring[ring_cols + rel] = E1out;
                    if (GAP == 2) my_ring[2 * ring_cols + rel] = E2out;
                }
                if (need_max) {
                    // per-lane candidate; columns past qlen only exist in vector qlen/PN and are masked there
                    const bool is_end = (v == end_sn);
                    int cand = (int)Hout;
                    if (is_end && end_sn == qlen / PN && col > qlen) cand = (int)inf;
                    if (sizeof(T) == 2) {
                        const unsigned key = ((unsigned)(cand + 32768) << 16) | ((unsigned)(PN - 1 - l) << 12) | ((unsigned)is_end << 11) | (unsigned)(2047 - v);
                        am_key = key > am_key ? key : am_key;
                    } else if (!am_any || (is_end ? cand >= am_val : cand > am_val)) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
                }
            }
            STAMP(3)
        }
        if (to_ring && lane == 0) S.b_rec[row % RB].w = row;      // H/E of this row are now readable from the score ring
        // ---- row arg-max, reference simd_abpoa_max_in_row :1043-1057 (tie-break: lowest lane, then the
        //      end_sn vector, then the lowest vector) and band hand-over :1059-1067
        int mx = d.inf_min, mi = -1;
        if (need_max && (b.dbg & 2)) { mx = 0; mi = imin(qlen, row + 1); }
        else if (need_max) {
            if (sizeof(T) == 2) {
                const unsigned kb = wave_max_u32(am_key);
                const int vmax = (int)(kb >> 16) - 32768;
                if (vmax > d.inf_min) { mx = vmax; mi = (2047 - (int)(kb & 0x7ff)) * PN + (PN - 1 - (int)((kb >> 12) & 0xf)); if (mi > qlen) mi = -1; }
            } else {
            int vmax = wave_max_i32(am_any ? am_val : INT_MIN);
            if (vmax > d.inf_min) {
                unsigned key = 0;
                if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                unsigned kb = wave_max_u32(key);
                int wl = PN - 1 - (int)(kb >> 27), wv = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu));
                mx = vmax; mi = wv * PN + wl;
                if (mi > qlen) mi = -1;          // cannot happen for a value above inf_min, kept for symmetry with qi[]
            }
            }
            if (b.want_trace && lane == 0) row_max_i[row] = mi;
            if (local) { if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; } }
            else if (extend) {
                if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; best_row_zd = row; }
                else if (b.zdrop > 0) {
                    int delta_index = gld_i32(row_remain + best_row_zd) - (qlen - rterm + remain_end + 1);
                    int dd = delta_index - (mi - best_j); if (dd < 0) dd = -dd;
                    if (best_score - mx > b.zdrop + (int)e1 * dd) break;
                }
            }
            if (banded) {
                const int out_i = mi + 1;
                const int o0 = fastmeta ? __builtin_amdgcn_readlane(tv_o[0], ti) : -1, o1 = fastmeta ? __builtin_amdgcn_readlane(tv_o[1], ti) : -1;
                if (fastmeta && o0 < lr_blk + RL && o1 < lr_blk + RL) {      // at most two successors, both inside the LDS window
                    int2 a2 = S.l_lr[(o0 >= 0 ? o0 : 0) % RL], b2 = S.l_lr[(o1 >= 0 ? o1 : 0) % RL];
                    a2.x = imin(a2.x, out_i); a2.y = imax(a2.y, out_i); b2.x = imin(b2.x, out_i); b2.y = imax(b2.y, out_i);
                    if (o0 >= 0 && lane == 0) S.l_lr[o0 % RL] = a2;
                    if (o1 >= 0 && lane == 0) S.l_lr[o1 % RL] = b2;
                } else {
                bool far = false;
                for (int t = lane; t < on; t += 64) {
                    const int tt = os + t - obase;
                    int o = S.t_out[tt < TP ? tt : 0];
                    if (tt >= TP) o = gld_i32(out_row + os + t);
                    if (o >= 0) {
                        if (o < lr_blk + RL) {
                            int2 v2 = S.l_lr[o % RL];
                            v2.x = imin(v2.x, out_i); v2.y = imax(v2.y, out_i);
                            S.l_lr[o % RL] = v2;
                        } else {                                       // beyond the LDS window: update the HBM copy
                            if (out_i > gld_i32(g_right + o)) g_right[o] = out_i;
                            if (out_i < gld_i32(g_left + o)) g_left[o] = out_i;
                            far = true;
                        }
                    }
                }
                if (__any(far)) { far_seen = true; asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
                }
            }
        } else if (b.want_trace && lane == 0) row_max_i[row] = -2;
        STAMP(4)
    }
    __syncthreads();
    if (tile_end > tile_beg && tile_beg + lane < tile_end && status == 0) {      // band geometry of the last (partial) tile
        const int r = tile_beg + lane;
        if (r <= last_done) { const int4 br = S.b_rec[r % RB]; g_bsn[r] = br.x; g_esn[r] = br.y; g_coff[r] = (long long)(uint32_t)br.z * PN; }
    }
    clk1 = (long long)__builtin_amdgcn_s_memtime();
    // ---- retire the left/right window to HBM (the arrays are in/out for the caller)
    if (banded && status == 0) {
        for (int i = lane; i < RL; i += 64) { const int r = lr_blk + i; if (r < gn) { const int2 v2 = S.l_lr[r % RL]; g_left[r] = v2.x; g_right[r] = v2.y; } }
    }
    (void)last_row;
#ifdef ABPOA_HIP_PROFILE
    for (int i_ = 0; i_ < 6; ++i_) seg_keep[i_] = seg[i_];
#endif
    }   // general row loop
    TailState ts; ts.cursor = cursor; ts.n_cells = n_cells; ts.status = status; ts.rows_done = rows_done; ts.best_score = best_score; ts.best_i = best_i; ts.best_j = best_j;
    ts.clk0 = clk0; ts.clk1 = clk1;
#ifdef ABPOA_HIP_PROFILE
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = seg_keep[i_];
#else
    for (int i_ = 0; i_ < 6; ++i_) ts.seg[i_] = 0;
#endif
    finish_alignment<T, GAP>(b, d, out_rec, ts);
}

}  // namespace abpoa_hip
