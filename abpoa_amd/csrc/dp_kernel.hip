// Adaptive-banded sequence-to-graph DP for gfx950 (MI355X).
//
// One 64-lane wavefront owns one alignment and walks its graph rows in topological order (rows of one
// alignment are strictly sequential: the band of row r depends on the arg-max of all its predecessor
// rows, reference src/simd_abpoa_align.c:1059-1067).  Lanes map to consecutive band columns, 64 columns
// ("chunk") at a time; inside a chunk the reference's SIMD register (pn = 16 int16 / 8 int32 lanes) is a
// group of pn adjacent lanes, so its whole-register lane shifts become DPP row shifts and its
// vector-to-vector carry ("first") is a wave-uniform scalar.  Score planes are stored band-compacted in
// HBM: row r owns P*(end_sn-beg_sn+1)*pn cells, written once with coalesced stores and re-read by
// successor rows and by the backtrack.
//
// Bit-exactness contract: every add/sub is done in the score width with
// two's-complement wrap, the masked log-step scan of SIMD_SET_F (:665-699) is reproduced step by step,
// and the band, arg-max tie-break and backtrack priority follow the reference literally.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <limits.h>
#include "engine.h"
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

#define OP_M   0x1
#define OP_E1  0x2
#define OP_E2  0x4
#define OP_E   0x6
#define OP_F1  0x8
#define OP_F2  0x10
#define OP_F   0x18
#define OP_ALL 0x1f

template <int CTRL>
__device__ __forceinline__ int dpp_mov(int old, int src) {
    return __builtin_amdgcn_update_dpp(old, src, CTRL, 0xF, 0xF, false);
}
// value of lane-S inside a 16-lane DPP row; lanes whose source falls outside the row keep `old`
template <int S>
__device__ __forceinline__ int row_shr(int old, int src) { return dpp_mov<0x110 + S>(old, src); }

__device__ __forceinline__ int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// wave-wide signed max; every lane must be active
__device__ __forceinline__ int wave_max_i32(int x) {
    x = imax(x, row_shr<1>(x, x));
    x = imax(x, row_shr<2>(x, x));
    x = imax(x, row_shr<4>(x, x));
    x = imax(x, row_shr<8>(x, x));
    int a = __builtin_amdgcn_readlane(x, 15), b = __builtin_amdgcn_readlane(x, 31);
    int c = __builtin_amdgcn_readlane(x, 47), d = __builtin_amdgcn_readlane(x, 63);
    return imax(imax(a, b), imax(c, d));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    auto umax = [](unsigned p, unsigned q) { return p > q ? p : q; };
    x = umax(x, (unsigned)row_shr<1>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<2>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<4>((int)x, (int)x));
    x = umax(x, (unsigned)row_shr<8>((int)x, (int)x));
    unsigned a = __builtin_amdgcn_readlane((int)x, 15), b = __builtin_amdgcn_readlane((int)x, 31);
    unsigned c = __builtin_amdgcn_readlane((int)x, 47), d = __builtin_amdgcn_readlane((int)x, 63);
    return umax(umax(a, b), umax(c, d));
}

template <typename T> struct Width;
template <> struct Width<int16_t> { static constexpr int PN = 16, LOGN = 4; };
template <> struct Width<int32_t> { static constexpr int PN = 8, LOGN = 3; };

// wrapping arithmetic in the score width (reference: _mm256_add/sub_epi16|32)
template <typename T> __device__ __forceinline__ T wadd(T a, T b) { return (T)((uint32_t)(int32_t)a + (uint32_t)(int32_t)b); }
template <typename T> __device__ __forceinline__ T wsub(T a, T b) { return (T)((uint32_t)(int32_t)a - (uint32_t)(int32_t)b); }
template <typename T> __device__ __forceinline__ T tmax(T a, T b) { return a > b ? a : b; }

// One step-by-step SIMD_SET_F (reference :665-699) on every pn-lane group of the wave at once.
// l = lane % PN.  set_num == PN selects the plain variant.
template <typename T>
__device__ __forceinline__ T set_f(T f, int l, int set_num, T e, T inf) {
    constexpr int PN = Width<T>::PN, LOGN = Width<T>::LOGN;
    T es = e; int cov = set_num;
#define SETF_STEP(K)                                                                          \
    if (K < LOGN) {                                                                           \
        constexpr int S = 1 << K;                                                             \
        if (K > 0) { es = wadd<T>(es, es); cov += S; }                                        \
        T t = wsub<T>(f, es);                                                                 \
        T sh = (T)row_shr<S>((int)inf, (int)t);                                               \
        if (PN == 8) sh = (l < S) ? inf : sh;      /* two vectors share a 16-lane DPP row */   \
        if (set_num != PN) sh = (l > cov) ? inf : sh;                                         \
        f = tmax<T>(f, sh);                                                                   \
    }
    SETF_STEP(0) SETF_STEP(1) SETF_STEP(2) SETF_STEP(3)
#undef SETF_STEP
    return f;
}

template <typename T>
struct RowMeta { int beg_sn, end_sn; long long off; };

// GAP: 0 linear, 1 affine, 2 convex (reference gap_mode)
template <typename T, int GAP>
__device__ void align_one(const DevBatch &b, const AlnDesc &d, AlnOut *out_rec) {
    constexpr int PN = Width<T>::PN, NV = 64 / PN;
    constexpr int P = GAP == 0 ? 1 : (GAP == 1 ? 3 : 5);
    constexpr int PL_E1 = 1, PL_E2 = 2, PL_F1 = GAP == 1 ? 2 : 3, PL_F2 = 4;
    const int lane = threadIdx.x & 63, l = lane % PN, vvl = lane / PN;
    const int gn = d.n_rows, qlen = d.qlen, m = b.m, w = d.w;
    const bool local = b.align_mode == ABPOA_HIP_LOCAL_MODE, extend = b.align_mode == ABPOA_HIP_EXTEND_MODE;
    const bool banded = b.wb >= 0;
    const T inf = (T)d.inf_min;
    const T e1 = (T)b.e1, o1 = (T)b.o1, oe1 = (T)(b.o1 + b.e1), e2 = (T)b.e2, o2 = (T)b.o2, oe2 = (T)(b.o2 + b.e2);
    const int dp_sn = (qlen + PN) / PN;

    const uint8_t *query = b.query + d.query_off;
    const uint8_t *row_base = b.row_base + d.row0;
    const int32_t *row_node_id = b.row_node_id + d.row0;
    const int32_t *row_remain = b.row_remain + d.row0;
    const uint8_t *row_active = b.row_active + d.row0;
    const int32_t *pred_off = b.pred_off + d.poff0, *pred_row = b.pred_row + d.pred0;
    const int32_t *out_off = b.out_off + d.poff0, *out_row = b.out_row + d.out0;
    int32_t *left = b.left + d.row0, *right = b.right + d.row0;
    int32_t *bsn = b.dp_beg_sn + d.row0, *esn = b.dp_end_sn + d.row0, *row_max_i = b.row_max_i + d.row0;
    int64_t *cell_off = b.row_cell_off + d.row0;
    T *planes = (T *)(b.planes + d.plane_off);
    const int32_t *mat = b.mat;

    // dp_end as the reference stores it: vector-rounded when banded and for row 0, qlen otherwise
    auto dp_end_of = [&](int row, int end_sn_row) { return (banded || row == 0) ? (end_sn_row + 1) * PN - 1 : qlen; };

    long long cursor = 0;          // next free arena cell
    long long n_cells = 0;
    int status = 0;

    // ------------------------------------------------------------------ row 0, reference :553-662
    {
        int dp_end0;
        if (banded) {
            if (lane == 0) { left[0] = 0; right[0] = 0; }
            for (int t = out_off[0] + lane; t < out_off[1]; t += 64) {
                int o = out_row[t];
                if (o >= 0 && row_active[o]) { left[o] = 1; right[o] = 1; }
            }
            int r = row_remain[0] - row_remain[gn - 1] - 1;
            dp_end0 = imin(qlen, imax(0, qlen - r) + w);          // right[0] == 0
        } else dp_end0 = qlen;
        int end_sn0 = dp_end0 / PN, W0 = (end_sn0 + 1) * PN;
        if ((long long)W0 * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; }
        else {
            if (lane == 0) { bsn[0] = 0; esn[0] = end_sn0; cell_off[0] = 0; row_max_i[0] = -2; }
            for (int i = lane; i < W0; i += 64) {
                if (local) { for (int p = 0; p < P; ++p) planes[(long long)p * W0 + i] = 0; }
                else if (GAP == 0) planes[i] = (T)(-(int)e1 * i);
                else if (GAP == 1) {
                    T g = (T)(-(int)o1 - (int)e1 * i);
                    planes[i] = i == 0 ? (T)0 : g;
                    planes[(long long)PL_E1 * W0 + i] = i == 0 ? (T)(-(int)oe1) : inf;
                    planes[(long long)PL_F1 * W0 + i] = i == 0 ? inf : g;
                } else {
                    T g1 = (T)(-(int)o1 - (int)e1 * i), g2 = (T)(-(int)o2 - (int)e2 * i);
                    planes[i] = i == 0 ? (T)0 : tmax<T>(g1, g2);
                    planes[(long long)PL_E1 * W0 + i] = i == 0 ? (T)(-(int)oe1) : inf;
                    planes[(long long)PL_E2 * W0 + i] = i == 0 ? (T)(-(int)oe2) : inf;
                    planes[(long long)PL_F1 * W0 + i] = i == 0 ? inf : g1;
                    planes[(long long)PL_F2 * W0 + i] = i == 0 ? inf : g2;
                }
            }
            cursor = (long long)W0 * P;
        }
    }
    __syncthreads();

    int best_score = d.inf_min, best_i = 0, best_j = 0, best_row_zd = 0;
    const int remain_end = (banded || b.zdrop > 0) ? row_remain[gn - 1] : 0;
    const bool need_max = local || extend || banded;

    // ------------------------------------------------------------------ rows 1 .. gn-2, reference :1105
    for (int row = 1; row < gn - 1 && status == 0; ++row) {
        if (!row_active[row]) { if (lane == 0) { bsn[row] = -1; esn[row] = -1; row_max_i[row] = -2; } continue; }
        const int base = row_base[row];
        const int ps = pred_off[row], np = pred_off[row + 1] - ps;
        int beg_sn, end_sn, max_pre_end_sn;
        if (!banded) { beg_sn = 0; end_sn = qlen / PN; max_pre_end_sn = end_sn; }        // reference :706-709
        else {                                                                          // reference :710-720
            int r = row_remain[row] - remain_end - 1;
            int beg = imax(0, imin(left[row], qlen - r) - w), end = imin(qlen, imax(right[row], qlen - r) + w);
            beg_sn = beg / PN; int min_pre_beg_sn = INT_MAX; max_pre_end_sn = -1;
            for (int k = 0; k < np; ++k) {
                int p = pred_row[ps + k];
                min_pre_beg_sn = imin(min_pre_beg_sn, bsn[p]); max_pre_end_sn = imax(max_pre_end_sn, esn[p]);
            }
            if (beg_sn < min_pre_beg_sn) beg_sn = min_pre_beg_sn;
            end_sn = end / PN;
        }
        const int Wr = (end_sn - beg_sn + 1) * PN;
        const long long off = cursor;
        if (off + (long long)Wr * P > d.plane_cap) { status = ABPOA_HIP_STATUS_OVERFLOW; break; }
        cursor += (long long)Wr * P;
        n_cells += Wr;
        if (lane == 0) { bsn[row] = beg_sn; esn[row] = end_sn; cell_off[row] = off; }
        T *H = planes + off;
        const int nchunk = (Wr + 63) >> 6;
        T first = 0, first2 = 0;
        // running arg-max state of this lane (reference :1043-1057)
        int am_val = INT_MIN, am_v = 0, am_isend = 0; bool am_any = false;

        for (int c = 0; c < nchunk; ++c) {
            const int rel = c * 64 + lane;
            const bool in_band = rel < Wr;
            const int col = beg_sn * PN + rel;
            const int v = beg_sn + c * NV + vvl;
            T Mv = inf, E1v = inf, E2v = inf;
            // query profile value, reference :504-510
            T q = 0;
            if (in_band && col >= 1 && col <= qlen) q = (T)mat[base * m + query[col - 1]];
            // ---- predecessors, reference :722-761 / :803-852 / :912-969
            for (int k = 0; k < np; ++k) {
                const int p = pred_row[ps + k];
                const int pb = bsn[p], pe = esn[p];
                const long long poff = cell_off[p];
                const int Wp = (pe - pb + 1) * PN;
                const T *Hp = planes + poff;
                const int p_stored_end = (pe + 1) * PN - 1;          // last stored column of the predecessor row
                int bs, es_h, es_e; T carry;
                if (local) { bs = 0; es_h = end_sn; es_e = end_sn; carry = 0; }
                else {
                    if (pb < beg_sn) { bs = beg_sn; carry = Hp[beg_sn * PN - 1 - pb * PN]; }
                    else { bs = pb; carry = inf; }
                    es_h = imin(imin((dp_end_of(p, pe) + 1) / PN, end_sn), dp_sn - 1);
                    es_e = imin(pe, end_sn);
                }
                const bool inH = in_band && v >= bs && v <= es_h;
                if (inH) {
                    T hval;
                    if (col == bs * PN) hval = carry;
                    else hval = (col - 1 <= p_stored_end) ? Hp[col - 1 - pb * PN] : inf;
                    if (GAP == 0) {
                        T vert = (col <= p_stored_end) ? Hp[col - pb * PN] : inf;
                        hval = tmax<T>(wadd<T>(hval, q), wsub<T>(vert, e1));
                    }
                    Mv = (k == 0) ? hval : tmax<T>(Mv, hval);
                }
                if (GAP != 0) {
                    const bool inE = in_band && v >= bs && v <= es_e;
                    if (inE) {
                        T ev = Hp[(long long)PL_E1 * Wp + col - pb * PN];
                        E1v = (k == 0) ? ev : tmax<T>(E1v, ev);
                        if (GAP == 2) {
                            T ev2 = Hp[(long long)PL_E2 * Wp + col - pb * PN];
                            E2v = (k == 0) ? ev2 : tmax<T>(E2v, ev2);
                        }
                    }
                }
            }
            // ---- in-row part
            T Hout, E1out = 0, E2out = 0, F1 = inf, F2 = inf;
            if (GAP == 0) {
                // reference :762-778
                T h = Mv;
                if (c == 0) first = (T)__builtin_amdgcn_readlane((int)h, 0);
#pragma unroll
                for (int vv = 0; vv < NV; ++vv) {
                    const int vg = beg_sn + c * NV + vv;
                    if (vg <= end_sn) {
                        int set_num = PN;
                        if (!local && vg > max_pre_end_sn) set_num = (vg == max_pre_end_sn + 1) ? 1 : 0;
                        T hv = tmax<T>(h, l == 0 ? first : inf);
                        hv = set_f<T>(hv, l, set_num, e1, inf);
                        if (vvl == vv) h = hv;
                        first = wsub<T>((T)__builtin_amdgcn_readlane((int)hv, vv * PN + PN - 1), e1);
                    }
                }
                Hout = local ? tmax<T>((T)0, h) : h;
            } else {
                T h = wadd<T>(Mv, q);                                   // reference :854-856 / :972-974
                T hs = h;                                               // value the F recurrence opens from
                if (GAP == 2) hs = tmax<T>(tmax<T>(h, E1v), E2v);       // reference :988
                if (c == 0) { first = (T)__builtin_amdgcn_readlane((int)h, 0); first2 = first; }   // :858 / :976-977
#pragma unroll
                for (int vv = 0; vv < NV; ++vv) {
                    const int vg = beg_sn + c * NV + vv;
                    if (vg <= end_sn) {
                        int set_num = PN;
                        if (!local && vg > max_pre_end_sn) set_num = (vg == max_pre_end_sn + 1) ? 2 : 1;
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
            if (in_band) {
                H[rel] = Hout;
                if (GAP != 0) {
                    H[(long long)PL_E1 * Wr + rel] = E1out;
                    H[(long long)PL_F1 * Wr + rel] = F1;
                    if (GAP == 2) { H[(long long)PL_E2 * Wr + rel] = E2out; H[(long long)PL_F2 * Wr + rel] = F2; }
                }
                if (need_max) {
                    // per-lane candidate; columns past qlen only exist in vector qlen/PN and are masked there
                    const bool is_end = (v == end_sn);
                    int cand = (int)Hout;
                    if (is_end && end_sn == qlen / PN && col > qlen) cand = (int)inf;
                    if (!am_any || (is_end ? cand >= am_val : cand > am_val)) { am_val = cand; am_v = v; am_isend = is_end; am_any = true; }
                }
            }
        }
        // ---- row arg-max, reference simd_abpoa_max_in_row :1043-1057 (tie-break: lowest lane, then the
        //      end_sn vector, then the lowest vector) and band hand-over :1059-1067
        int mx = d.inf_min, mi = -1;
        if (need_max) {
            int vmax = wave_max_i32(am_any ? am_val : INT_MIN);
            if (vmax > d.inf_min) {
                unsigned key = 0;
                if (am_any && am_val == vmax) key = ((unsigned)(PN - 1 - l) << 27) | ((unsigned)am_isend << 26) | (0x3FFFFFFu - (unsigned)am_v);
                unsigned kb = wave_max_u32(key);
                int wl = PN - 1 - (int)(kb >> 27), wv = (int)(0x3FFFFFFu - (kb & 0x3FFFFFFu));
                mx = vmax; mi = wv * PN + wl;
                if (mi > qlen) mi = -1;          // cannot happen for a value above inf_min, kept for symmetry with qi[]
            }
            if (lane == 0) row_max_i[row] = mi;
            if (local) { if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; } }
            else if (extend) {
                if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; best_row_zd = row; }
                else if (b.zdrop > 0) {
                    int delta_index = row_remain[best_row_zd] - row_remain[row];
                    int dd = delta_index - (mi - best_j); if (dd < 0) dd = -dd;
                    if (best_score - mx > b.zdrop + (int)e1 * dd) { __syncthreads(); break; }
                }
            }
            if (banded) {
                const int out_i = mi + 1;
                for (int t = out_off[row] + lane; t < out_off[row + 1]; t += 64) {
                    int o = out_row[t];
                    if (o >= 0) {
                        if (out_i > right[o]) right[o] = out_i;
                        if (out_i < left[o]) left[o] = out_i;
                    }
                }
            }
        } else if (lane == 0) row_max_i[row] = -2;
        __syncthreads();    // this wave's stores (planes, band, left/right) before the next row's loads
    }

    // ------------------------------------------------------------------ global best, reference :1028-1041
    if (status == 0 && b.align_mode == ABPOA_HIP_GLOBAL_MODE) {
        for (int k = pred_off[gn - 1]; k < pred_off[gn]; ++k) {
            int in_row = pred_row[k];
            int pe = esn[in_row], pb = bsn[in_row];
            int dpe = dp_end_of(in_row, pe);
            int end = qlen > dpe ? dpe : qlen;
            int score = (int)planes[cell_off[in_row] + end - pb * PN];
            if (score > best_score) { best_score = score; best_i = in_row; best_j = end; }
        }
    }

    // ------------------------------------------------------------------ backtrack, reference :109-429
    int n_cigar = 0, node_s = 0, node_e = 0, query_s = 0, query_e = 0, n_aln = 0, n_match = 0;
    if (status == 0 && b.ret_cigar && lane == 0) {
        uint64_t *cg = b.cigar + d.cigar_off;
        const int cap = d.cigar_cap;
        auto push = [&](int op, int len, int node_id, int query_id) {      // reference abpoa_align.h:54-73
            uint64_t L = (uint64_t)(int64_t)len;
            if (n_cigar == 0 || op != ABPOA_HIP_CINS || op != (int)(cg[n_cigar - 1] & 0xf)) {
                if (n_cigar >= cap) { status = ABPOA_HIP_EBACKTRACK; return; }
                uint64_t n_id = (uint64_t)(int64_t)node_id, q_id = (uint64_t)(int64_t)query_id;
                if (op == ABPOA_HIP_CMATCH) cg[n_cigar++] = n_id << 34 | q_id << 4 | (uint64_t)op;
                else if (op == ABPOA_HIP_CINS) cg[n_cigar++] = q_id << 34 | L << 4 | (uint64_t)op;
                else cg[n_cigar++] = n_id << 34 | L << 4 | (uint64_t)op;
            } else cg[n_cigar - 1] += L << 4;
        };
        // cell (row, plane, col) with the row's band geometry
        auto cell = [&](int row_, int plane, int col_) -> int {
            int pb = bsn[row_], pe = esn[row_];
            long long Wp = (long long)(pe - pb + 1) * PN;
            return (int)planes[cell_off[row_] + plane * Wp + (col_ - pb * PN)];
        };
        auto in_range = [&](int row_, int col_) {                            // dp_beg <= col <= dp_end
            int pb = bsn[row_], pe = esn[row_];
            return col_ >= pb * PN && col_ <= dp_end_of(row_, pe);
        };
        auto stored = [&](int row_, int col_) {                              // column physically stored
            int pb = bsn[row_], pe = esn[row_];
            return col_ >= pb * PN && col_ <= (pe + 1) * PN - 1;
        };
        int i = best_i, j = best_j, start_i = best_i, start_j = best_j, cur_op = OP_ALL, indel_first = 1;
        if (best_j < qlen) push(ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
        while (i > 0 && j > 0 && status == 0) {
            const int Hij = cell(i, 0, j);
            if (local && Hij == 0) break;
            start_i = i; start_j = j;
            const int ps = pred_off[i], np = pred_off[i + 1] - ps;
            const int id = row_node_id[i];
            const int s = mat[m * row_base[i] + query[j - 1]];
            const int is_match = row_base[i] == query[j - 1];
            int hit = 0;
            auto try_match = [&](int set_indel) {
                for (int k = 0; k < np; ++k) {
                    int pr = pred_row[ps + k];
                    if (!in_range(pr, j - 1)) continue;
                    if (cell(pr, 0, j - 1) + s == Hij) {
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
                        int pr = pred_row[ps + k];
                        if (!in_range(pr, j)) continue;
                        if (cell(pr, 0, j) - (int)e1 == Hij) { push(ABPOA_HIP_CDEL, 1, id, j - 1); i = pr; hit = 1; break; }
                    }
                }
                if (!hit && stored(i, j - 1) && cell(i, 0, j - 1) - (int)e1 == Hij) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; hit = 1; ++n_aln; }
                if (!hit && indel_first == 1) try_match(1);
            } else {
                if ((cur_op & OP_M) && indel_first == 0) try_match(0);
                if (!hit && (cur_op & OP_E)) {
                    for (int k = 0; k < np && !hit; ++k) {
                        int pr = pred_row[ps + k];
                        if (!in_range(pr, j)) continue;
                        for (int x = 1; x <= (GAP == 2 ? 2 : 1); ++x) {
                            const int bit = x == 1 ? OP_E1 : OP_E2, pl = x == 1 ? PL_E1 : PL_E2;
                            const int ex = x == 1 ? (int)e1 : (int)e2, oex = x == 1 ? (int)oe1 : (int)oe2;
                            if (!(cur_op & bit)) continue;
                            const int preE = cell(pr, pl, j);
                            const bool ok = (cur_op & OP_M) ? (Hij == preE) : (cell(i, pl, j) == preE - ex);
                            if (ok) {
                                cur_op = (cell(pr, 0, j) - oex == preE) ? (OP_M | OP_F) : bit;
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
                        const int Fij = cell(i, pl, j);
                        if (!(cur_op & OP_M) || Hij == Fij) {
                            if (stored(i, j - 1)) {
                                if (cell(i, 0, j - 1) - oex == Fij) { cur_op = OP_M | OP_E; hit = 1; }
                                else if (cell(i, pl, j - 1) - ex == Fij) { cur_op = bit; hit = 1; }
                            }
                        }
                    }
                    if (hit) { push(ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
                }
                if (!hit && (cur_op & OP_M) && indel_first == 1) try_match(1);
            }
            if (!hit && status == 0) status = ABPOA_HIP_EBACKTRACK;
        }
        if (status == 0) {
            if (j > 0) push(ABPOA_HIP_CINS, j, -1, j - 1);
            if (!b.rev_cigar) for (int k = 0; k < n_cigar >> 1; ++k) { uint64_t t = cg[k]; cg[k] = cg[n_cigar - 1 - k]; cg[n_cigar - 1 - k] = t; }
            node_e = row_node_id[best_i]; query_e = best_j - 1;
            node_s = row_node_id[start_i]; query_s = start_j - 1;
        }
    }
    if (lane == 0) {
        AlnOut o;
        o.status = status; o.best_score = best_score; o.best_row = best_i; o.best_col = best_j;
        o.node_s = node_s; o.node_e = node_e; o.query_s = query_s; o.query_e = query_e;
        o.n_aln_bases = n_aln; o.n_matched_bases = n_match; o.n_cigar = n_cigar; o.pad = 0;
        o.n_cells = n_cells; o.cells_used = cursor;
        *out_rec = o;
    }
}

template <int GAP>
__global__ void __launch_bounds__(64) dp_kernel(const DevBatch b) {
    const int a = blockIdx.x;
    if (a >= b.n) return;
    const AlnDesc d = b.aln[a];
    if (d.bits == 16) align_one<int16_t, GAP>(b, d, b.out + a);
    else align_one<int32_t, GAP>(b, d, b.out + a);
}

hipError_t launch_dp(const DevBatch &b, hipStream_t stream) {
    if (b.n <= 0) return hipSuccess;
    dim3 grid(b.n), block(64);
    switch (b.gap_mode) {
        case ABPOA_HIP_LINEAR_GAP: hipLaunchKernelGGL(dp_kernel<0>, grid, block, 0, stream, b); break;
        case ABPOA_HIP_AFFINE_GAP: hipLaunchKernelGGL(dp_kernel<1>, grid, block, 0, stream, b); break;
        default: hipLaunchKernelGGL(dp_kernel<2>, grid, block, 0, stream, b); break;
    }
    return hipGetLastError();
}

}  // namespace abpoa_hip
