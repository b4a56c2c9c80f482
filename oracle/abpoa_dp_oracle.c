/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH (see abpoa_dp_oracle.h).
 *
 * Scalar restatement of abPOA v1.4.1's adaptive-banded sequence-to-graph DP.  The reference computes
 * with AVX2 registers of pn = 16 (int16) or 8 (int32) lanes; here every "virtual vector" is a loop
 * over pn array elements, and every add/sub wraps to the score width exactly like
 * _mm256_add/sub_epi16/32.  All `ref:` citations are relative to /root/reference/src/.
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <limits.h>
#include "abpoa_dp_oracle.h"

#define OP_M   0x1   /* ref: abpoa_align.h:20-27 */
#define OP_E1  0x2
#define OP_E2  0x4
#define OP_E   0x6
#define OP_F1  0x8
#define OP_F2  0x10
#define OP_F   0x18
#define OP_ALL 0x1f

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin3(int a, int b, int c) { return imin(a, imin(b, c)); }
static inline int imax3(int a, int b, int c) { return imax(a, imax(b, c)); }

typedef struct {
    int bits, pn, log_n, dp_sn, width, P, inf, local;
    int qlen, gn, m;
    int o1, e1, oe1, o2, e2, oe2;
    int32_t *planes;         /* [gn][P][width] */
    int32_t *qp;             /* [m][width] query profile, ref: simd_abpoa_align.c:502-510 */
    int *dp_beg, *dp_end, *dp_beg_sn, *dp_end_sn;
} ctx_t;

/* wrap to the score width (two's complement), ref: SIMDAddi16/SIMDSubi16 = _mm256_add/sub_epi16 */
static inline int W(const ctx_t *c, long long x) {
    return c->bits == 16 ? (int)(int16_t)(uint16_t)(uint64_t)x : (int)(int32_t)(uint32_t)(uint64_t)x;
}
static inline int32_t *PL(const ctx_t *c, int row, int plane) {
    return c->planes + ((int64_t)row * c->P + plane) * c->width;
}

/* ref: simd_abpoa_align.c:1672-1683 */
int abpoa_oracle_score_bits(const abpoa_hip_scoring_t *sc, int n_rows, int qlen, int32_t *inf_min) {
    int oe1 = sc->gap_open1 + sc->gap_ext1, oe2 = sc->gap_open2 + sc->gap_ext2;
    int len = qlen > n_rows ? qlen : n_rows;
    int max_score = imax(qlen * sc->max_mat, len * sc->gap_ext1 + sc->gap_open1);
    int bits, lo;
    if (max_score <= INT16_MAX - sc->min_mis - oe1 - oe2) { bits = 16; lo = INT16_MIN; }
    else { bits = 32; lo = INT32_MIN; }
    if (inf_min) *inf_min = imax3(lo + sc->min_mis, lo + oe1, lo + oe2) + 31 * imax(sc->gap_ext1, sc->gap_ext2);
    return bits;
}

/* The masked log-step scan, ref: SIMD_SET_F simd_abpoa_align.c:665-699.  F has pn lanes. */
static void set_f(const ctx_t *c, int *F, int set_num, int e) {
    int pn = c->pn, k, l, cov = set_num, sh[16], es = e; /* es = GAP_ExS[k], doubled with a wrapping add (:471-475) */
    for (k = 0; k < c->log_n; ++k) {
        int s = 1 << k;
        if (k > 0) { es = W(c, (long long)es + es); cov += s; }
        for (l = 0; l < pn; ++l) {
            if (l < s) sh[l] = c->inf;                       /* zero-filled shift | PRE_MIN[s] */
            else if (set_num != pn && l > cov) sh[l] = c->inf; /* & PRE_MASK[cov] | SUF_MIN[cov] */
            else sh[l] = W(c, (long long)F[l - s] - es);
        }
        for (l = 0; l < pn; ++l) F[l] = imax(F[l], sh[l]);
    }
}

/* ref: GET_AD_DP_BEGIN/END abpoa_align.h:34-35 */
static int ad_beg(const abpoa_hip_problem_t *p, int row, int w) {
    int r = p->row_remain[row] - p->row_remain[p->n_rows - 1] - 1;
    return imax(0, imin(p->max_pos_left[row], p->qlen - r) - w);
}
static int ad_end(const abpoa_hip_problem_t *p, int row, int w) {
    int r = p->row_remain[row] - p->row_remain[p->n_rows - 1] - 1;
    return imin(p->qlen, imax(p->max_pos_right[row], p->qlen - r) + w);
}

/* Row 0, ref: simd_abpoa_{lg,ag,cg}_first_row / _first_dp simd_abpoa_align.c:553-662 */
static void first_row(ctx_t *c, const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, int w) {
    int pn = c->pn, i, pl, end_sn_fill;
    if (sc->wb >= 0) {
        p->max_pos_left[0] = p->max_pos_right[0] = 0;
        for (i = p->out_off[0]; i < p->out_off[1]; ++i) {
            int o = p->out_row[i];
            if (o >= 0 && (p->row_active == NULL || p->row_active[o]))
                p->max_pos_left[o] = p->max_pos_right[o] = 1;
        }
        c->dp_beg[0] = 0; c->dp_end[0] = ad_end(p, 0, w);
    } else { c->dp_beg[0] = 0; c->dp_end[0] = c->qlen; }
    c->dp_beg_sn[0] = 0; c->dp_end_sn[0] = c->dp_end[0] / pn;
    c->dp_beg[0] = 0; c->dp_end[0] = (c->dp_end_sn[0] + 1) * pn - 1;
    end_sn_fill = imin(c->dp_end_sn[0] + 1, c->dp_sn - 1);
    int32_t *H = PL(c, 0, 0);
    if (c->local) {
        for (pl = 0; pl < c->P; ++pl) { int32_t *x = PL(c, 0, pl); for (i = 0; i < (end_sn_fill + 1) * pn; ++i) x[i] = 0; }
        return;
    }
    int nE = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 0 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 1 : 2);
    for (pl = 0; pl <= nE; ++pl) { int32_t *x = PL(c, 0, pl); for (i = 0; i < (end_sn_fill + 1) * pn; ++i) x[i] = c->inf; }
    if (sc->gap_mode == ABPOA_HIP_LINEAR_GAP) {
        for (i = 0; i <= c->dp_end[0]; ++i) H[i] = W(c, -(long long)c->e1 * i);
    } else if (sc->gap_mode == ABPOA_HIP_AFFINE_GAP) {
        int32_t *E1 = PL(c, 0, 1), *F1 = PL(c, 0, 2);
        H[0] = 0; E1[0] = W(c, -(long long)c->oe1); F1[0] = c->inf;
        for (i = 1; i <= c->dp_end[0]; ++i) F1[i] = H[i] = W(c, -(long long)c->o1 - (long long)c->e1 * i);
    } else {
        int32_t *E1 = PL(c, 0, 1), *E2 = PL(c, 0, 2), *F1 = PL(c, 0, 3), *F2 = PL(c, 0, 4);
        H[0] = 0; E1[0] = W(c, -(long long)c->oe1); E2[0] = W(c, -(long long)c->oe2); F1[0] = F2[0] = c->inf;
        for (i = 1; i <= c->dp_end[0]; ++i) {
            F1[i] = W(c, -(long long)c->o1 - (long long)c->e1 * i);
            F2[i] = W(c, -(long long)c->o2 - (long long)c->e2 * i);
            H[i] = imax(F1[i], F2[i]);
        }
    }
}

/* One DP row, ref: simd_abpoa_lg_dp :701-779, simd_abpoa_ag_dp :781-885, simd_abpoa_cg_dp :887-1010 */
static void dp_row(ctx_t *c, const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, int row, int w) {
    const int pn = c->pn, qlen = c->qlen, gm = sc->gap_mode, inf = c->inf;
    const int np = p->pred_off[row + 1] - p->pred_off[row];
    const int *preds = p->pred_row + p->pred_off[row];
    const int32_t *q = c->qp + (int64_t)p->row_base[row] * c->width;
    int beg, end, beg_sn, end_sn, min_pre_beg_sn, max_pre_end_sn, k, v, l, i;
    /* band, ref :706-720 */
    if (sc->wb < 0) {
        beg = c->dp_beg[row] = 0; end = c->dp_end[row] = qlen;
        beg_sn = c->dp_beg_sn[row] = 0; end_sn = c->dp_end_sn[row] = qlen / pn;
        min_pre_beg_sn = 0; max_pre_end_sn = end_sn;
    } else {
        beg = ad_beg(p, row, w); end = ad_end(p, row, w);
        beg_sn = beg / pn; min_pre_beg_sn = INT_MAX; max_pre_end_sn = -1;
        for (k = 0; k < np; ++k) {
            min_pre_beg_sn = imin(min_pre_beg_sn, c->dp_beg_sn[preds[k]]);
            max_pre_end_sn = imax(max_pre_end_sn, c->dp_end_sn[preds[k]]);
        }
        if (beg_sn < min_pre_beg_sn) beg_sn = min_pre_beg_sn;
        c->dp_beg_sn[row] = beg_sn; beg = c->dp_beg[row] = beg_sn * pn;
        end_sn = c->dp_end_sn[row] = end / pn; end = c->dp_end[row] = (end_sn + 1) * pn - 1;
    }
    (void)beg; (void)end;
    int32_t *H = PL(c, row, 0);
    int32_t *E1 = gm != ABPOA_HIP_LINEAR_GAP ? PL(c, row, 1) : NULL;
    int32_t *E2 = gm == ABPOA_HIP_CONVEX_GAP ? PL(c, row, 2) : NULL;
    int32_t *F1 = gm == ABPOA_HIP_AFFINE_GAP ? PL(c, row, 2) : (gm == ABPOA_HIP_CONVEX_GAP ? PL(c, row, 3) : NULL);
    int32_t *F2 = gm == ABPOA_HIP_CONVEX_GAP ? PL(c, row, 4) : NULL;

    /* predecessors, ref :722-761 / :803-852 / :912-969.  _beg_sn/_end_sn persist across the loop exactly
     * as the reference's locals do (local mode never re-assigns them for k>0 in the H part). */
    int _beg_sn = 0, _end_sn = 0;
    for (k = 0; k < np; ++k) {
        int pr = preds[k];
        const int32_t *pH = PL(c, pr, 0);
        const int32_t *pE1 = E1 ? PL(c, pr, 1) : NULL, *pE2 = E2 ? PL(c, pr, 2) : NULL;
        int pre_end = c->dp_end[pr], pre_beg_sn = c->dp_beg_sn[pr], pre_end_sn = c->dp_end_sn[pr];
        int first;
        if (c->local) {
            if (k == 0) { _beg_sn = 0; _end_sn = end_sn; }
            first = 0;
        } else {
            if (pre_beg_sn < beg_sn) { _beg_sn = beg_sn; first = pH[beg_sn * pn - 1]; }
            else { _beg_sn = pre_beg_sn; first = inf; }
            _end_sn = imin3((pre_end + 1) / pn, end_sn, c->dp_sn - 1);
            if (k == 0) {
                for (v = beg_sn; v < _beg_sn; ++v) for (l = 0; l < pn; ++l) H[v * pn + l] = inf;
                for (v = _end_sn + 1; v <= imin(end_sn + 1, c->dp_sn - 1); ++v) for (l = 0; l < pn; ++l) H[v * pn + l] = inf;
            }
        }
        for (v = _beg_sn; v <= _end_sn; ++v) {
            for (l = 0; l < pn; ++l) {
                int j = v * pn + l;
                int m_in = (l == 0) ? first : pH[j - 1];      /* first | (pre_dp_h << 1 lane) */
                int val;
                if (gm == ABPOA_HIP_LINEAR_GAP) {
                    val = imax(W(c, (long long)m_in + q[j]), W(c, (long long)pH[j] - c->e1));
                } else val = m_in;
                H[j] = (k == 0) ? val : imax(val, H[j]);
            }
            first = pH[v * pn + pn - 1];
        }
        if (gm != ABPOA_HIP_LINEAR_GAP) {
            if (!c->local) {
                _end_sn = imin(pre_end_sn, end_sn);
                if (k == 0) {
                    for (v = beg_sn; v < _beg_sn; ++v) for (l = 0; l < pn; ++l) { E1[v * pn + l] = inf; if (E2) E2[v * pn + l] = inf; }
                    for (v = _end_sn + 1; v <= end_sn; ++v) for (l = 0; l < pn; ++l) { E1[v * pn + l] = inf; if (E2) E2[v * pn + l] = inf; }
                }
            } else if (k > 0) _end_sn = imin(pre_end_sn, end_sn);   /* ref :849 / :963 (unconditional for k>0) */
            for (v = _beg_sn; v <= _end_sn; ++v) for (l = 0; l < pn; ++l) {
                int j = v * pn + l;
                E1[j] = (k == 0) ? pE1[j] : imax(pE1[j], E1[j]);
                if (E2) E2[j] = (k == 0) ? pE2[j] : imax(pE2[j], E2[j]);
            }
        }
    }

    int hv[16], f1[16], f2[16];
    if (gm == ABPOA_HIP_LINEAR_GAP) {
        /* ref :762-778 */
        int first = H[beg_sn * pn];
        for (v = beg_sn; v <= end_sn; ++v) {
            int set_num;
            if (c->local) set_num = pn;
            else if (v > max_pre_end_sn) set_num = (v == max_pre_end_sn + 1) ? 1 : 0;
            else set_num = pn;
            for (l = 0; l < pn; ++l) hv[l] = imax(H[v * pn + l], l == 0 ? first : inf);
            set_f(c, hv, set_num, c->e1);
            for (l = 0; l < pn; ++l) H[v * pn + l] = hv[l];
            first = W(c, (long long)hv[pn - 1] - c->e1);
        }
        if (c->local) for (i = 0; i < (end_sn + 1) * pn; ++i) H[i] = imax(0, H[i]);
        return;
    }
    /* ref :854-856 / :972-974 */
    for (i = beg_sn * pn; i < (end_sn + 1) * pn; ++i) H[i] = W(c, (long long)H[i] + q[i]);
    int first = H[beg_sn * pn], first2 = first;            /* ref :858 / :976-977 */
    for (v = beg_sn; v <= end_sn; ++v) {
        int set_num;
        if (c->local) set_num = pn;
        else if (v > max_pre_end_sn) set_num = (v == max_pre_end_sn + 1) ? 2 : 1;
        else set_num = pn;
        int32_t *h = H + v * pn, *e1 = E1 + v * pn, *e2 = E2 ? E2 + v * pn : NULL;
        if (gm == ABPOA_HIP_AFFINE_GAP) {
            /* ref :870-883 */
            for (l = 0; l < pn; ++l) f1[l] = W(c, (long long)(l == 0 ? first : h[l - 1]) - c->oe1);
            set_f(c, f1, set_num, c->e1);
            first = imax(h[pn - 1], W(c, (long long)f1[pn - 1] + c->o1));
            for (l = 0; l < pn; ++l) {
                int tmp = imax(h[l], e1[l]);
                int hh = imax(tmp, f1[l]);
                if (c->local) hh = imax(0, hh);
                int enew = imax(W(c, (long long)e1[l] - c->e1), W(c, (long long)hh - c->oe1));
                e1[l] = (hh == tmp) ? enew : (c->local ? 0 : inf);
                h[l] = hh; F1[v * pn + l] = f1[l];
            }
        } else {
            /* ref :987-1008 */
            for (l = 0; l < pn; ++l) hv[l] = imax(imax(h[l], e1[l]), e2[l]);
            for (l = 0; l < pn; ++l) {
                f1[l] = W(c, (long long)(l == 0 ? first : hv[l - 1]) - c->oe1);
                f2[l] = W(c, (long long)(l == 0 ? first2 : hv[l - 1]) - c->oe2);
            }
            set_f(c, f1, set_num, c->e1);
            set_f(c, f2, set_num, c->e2);
            first = imax(hv[pn - 1], W(c, (long long)f1[pn - 1] + c->o1));
            first2 = imax(hv[pn - 1], W(c, (long long)f2[pn - 1] + c->o2));
            for (l = 0; l < pn; ++l) {
                int hh = imax(hv[l], imax(f1[l], f2[l]));
                if (c->local) hh = imax(0, hh);
                int en1 = imax(W(c, (long long)e1[l] - c->e1), W(c, (long long)hh - c->oe1));
                int en2 = imax(W(c, (long long)e2[l] - c->e2), W(c, (long long)hh - c->oe2));
                if (c->local) { en1 = imax(0, en1); en2 = imax(0, en2); }
                h[l] = hh; e1[l] = en1; e2[l] = en2;
                F1[v * pn + l] = f1[l]; F2[v * pn + l] = f2[l];
            }
        }
    }
}

/* ref: simd_abpoa_max_in_row :1043-1057.  qi[j] = j for j<=qlen else -1 (:511-515). */
static void max_in_row(const ctx_t *c, int row, int *max_out, int *max_i_out) {
    int pn = c->pn, beg_sn = c->dp_beg_sn[row], end_sn = c->dp_end_sn[row], l, v;
    const int32_t *H = PL(c, row, 0);
    int a[16], b[16];
    for (l = 0; l < pn; ++l) {
        int j = end_sn * pn + l;
        a[l] = H[j]; b[l] = j <= c->qlen ? j : -1;
        if (end_sn == c->qlen / pn && 0 > b[l]) a[l] = c->inf;
    }
    for (v = beg_sn; v < end_sn; ++v) for (l = 0; l < pn; ++l) {
        int j = v * pn + l;
        if (H[j] > a[l]) { a[l] = H[j]; b[l] = j <= c->qlen ? j : -1; }
    }
    int mx = c->inf, mi = -1;
    for (l = 0; l < pn; ++l) if (a[l] > mx) { mx = a[l]; mi = b[l]; }
    *max_out = mx; *max_i_out = mi;
}

typedef struct { uint64_t *a; int n, m; } cig_t;
/* ref: abpoa_push_cigar abpoa_align.h:54-73 */
static void push_cigar(cig_t *cg, int op, int len, int32_t node_id, int32_t query_id) {
    uint64_t l = (uint64_t)(int64_t)len;
    if (cg->n == 0 || op != ABPOA_HIP_CINS || op != (int)(cg->a[cg->n - 1] & 0xf)) {
        if (cg->n == cg->m) { cg->m = cg->m ? cg->m << 1 : 4; cg->a = (uint64_t *)realloc(cg->a, (size_t)cg->m * sizeof(uint64_t)); }
        uint64_t n_id = (uint64_t)(int64_t)node_id, q_id = (uint64_t)(int64_t)query_id;
        if (op == ABPOA_HIP_CMATCH) cg->a[cg->n++] = n_id << 34 | q_id << 4 | (uint64_t)op;
        else if (op == ABPOA_HIP_CINS) cg->a[cg->n++] = q_id << 34 | l << 4 | (uint64_t)op;
        else cg->a[cg->n++] = n_id << 34 | l << 4 | (uint64_t)op;
    } else cg->a[cg->n - 1] += l << 4;
}

/* ref: simd_abpoa_{lg,ag,cg}_backtrack :109-429 */
static int backtrack(const ctx_t *c, const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p,
                     int best_i, int best_j, abpoa_hip_result_t *res) {
    const int gm = sc->gap_mode, qlen = c->qlen, m = c->m;
    const int pE1 = 1, pE2 = 2, pF1 = gm == ABPOA_HIP_AFFINE_GAP ? 2 : 3, pF2 = 4;
    cig_t cg = {0, 0, 0};
    int i = best_i, j = best_j, start_i = best_i, start_j = best_j, k, hit, cur_op = OP_ALL, indel_first = 1;
    int n_aln = 0, n_match = 0;
    if (best_j < qlen) push_cigar(&cg, ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
    while (i > 0 && j > 0) {
        const int32_t *H = PL(c, i, 0);
        if (c->local && H[j] == 0) break;
        start_i = i; start_j = j;
        const int np = p->pred_off[i + 1] - p->pred_off[i];
        const int *preds = p->pred_row + p->pred_off[i];
        int id = p->row_node_id[i];
        int s = sc->mat[m * p->row_base[i] + p->query[j - 1]];
        int is_match = p->row_base[i] == p->query[j - 1];
        hit = 0;
#define TRY_MATCH(set_indel) do {                                                              \
        for (k = 0; k < np; ++k) {                                                             \
            int pr = preds[k];                                                                 \
            if (j - 1 < c->dp_beg[pr] || j - 1 > c->dp_end[pr]) continue;                      \
            if (PL(c, pr, 0)[j - 1] + s == H[j]) {                                             \
                cur_op = OP_ALL; hit = 1;                                                      \
                push_cigar(&cg, ABPOA_HIP_CMATCH, 1, id, j - 1);                               \
                i = pr; --j; ++n_aln; n_match += is_match ? 1 : 0;                             \
                if (set_indel) indel_first = 0;                                                \
                break;                                                                         \
            }                                                                                  \
        } } while (0)
        if (gm == ABPOA_HIP_LINEAR_GAP) {
            if (indel_first == 0) TRY_MATCH(0);
            if (hit == 0) {
                for (k = 0; k < np; ++k) {
                    int pr = preds[k];
                    if (j < c->dp_beg[pr] || j > c->dp_end[pr]) continue;
                    if (PL(c, pr, 0)[j] - c->e1 == H[j]) {
                        push_cigar(&cg, ABPOA_HIP_CDEL, 1, id, j - 1);
                        i = pr; hit = 1; break;
                    }
                }
            }
            if (hit == 0 && H[j - 1] - c->e1 == H[j]) {
                push_cigar(&cg, ABPOA_HIP_CINS, 1, id, j - 1); j--; hit = 1; ++n_aln;
            }
            if (hit == 0 && indel_first == 1) TRY_MATCH(1);
        } else {
            if ((cur_op & OP_M) && indel_first == 0) TRY_MATCH(0);
            if (hit == 0 && (cur_op & OP_E)) {
                for (k = 0; k < np && !hit; ++k) {
                    int pr = preds[k], x;
                    if (j < c->dp_beg[pr] || j > c->dp_end[pr]) continue;
                    for (x = 1; x <= (gm == ABPOA_HIP_CONVEX_GAP ? 2 : 1); ++x) {
                        int bit = x == 1 ? OP_E1 : OP_E2, pl = x == 1 ? pE1 : pE2;
                        int ex = x == 1 ? c->e1 : c->e2, oex = x == 1 ? c->oe1 : c->oe2;
                        if (!(cur_op & bit)) continue;
                        const int32_t *preE = PL(c, pr, pl);
                        int ok = (cur_op & OP_M) ? (H[j] == preE[j]) : (PL(c, i, pl)[j] == preE[j] - ex);
                        if (ok) {
                            cur_op = (PL(c, pr, 0)[j] - oex == preE[j]) ? (OP_M | OP_F) : bit;
                            hit = 1; push_cigar(&cg, ABPOA_HIP_CDEL, 1, id, j - 1);
                            i = pr; break;
                        }
                    }
                }
            }
            if (hit == 0 && (cur_op & OP_F)) {
                int x;
                for (x = 1; x <= (gm == ABPOA_HIP_CONVEX_GAP ? 2 : 1) && !hit; ++x) {
                    int bit = x == 1 ? OP_F1 : OP_F2, pl = x == 1 ? pF1 : pF2;
                    int ex = x == 1 ? c->e1 : c->e2, oex = x == 1 ? c->oe1 : c->oe2;
                    if (!(cur_op & bit)) continue;
                    const int32_t *F = PL(c, i, pl);
                    if (!(cur_op & OP_M) || H[j] == F[j]) {
                        if (H[j - 1] - oex == F[j]) { cur_op = OP_M | OP_E; hit = 1; }
                        else if (F[j - 1] - ex == F[j]) { cur_op = bit; hit = 1; }
                    }
                }
                if (hit) { push_cigar(&cg, ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
            }
            if (hit == 0 && (cur_op & OP_M) && indel_first == 1) TRY_MATCH(1);
        }
#undef TRY_MATCH
        if (hit == 0) { free(cg.a); res->status = ABPOA_HIP_EBACKTRACK; return ABPOA_HIP_EBACKTRACK; }
    }
    if (j > 0) push_cigar(&cg, ABPOA_HIP_CINS, j, -1, j - 1);
    if (!sc->rev_cigar) { /* ref: abpoa_reverse_cigar abpoa_align.h:88-96 */
        for (k = 0; k < cg.n >> 1; ++k) { uint64_t t = cg.a[k]; cg.a[k] = cg.a[cg.n - 1 - k]; cg.a[cg.n - 1 - k] = t; }
    }
    res->cigar = cg.a; res->n_cigar = cg.n;
    res->node_e = p->row_node_id[best_i]; res->query_e = best_j - 1;
    res->node_s = p->row_node_id[start_i]; res->query_s = start_j - 1;
    res->n_aln_bases = n_aln; res->n_matched_bases = n_match;
    return 0;
}

int abpoa_oracle_align(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p,
                       abpoa_hip_result_t *res, abpoa_oracle_trace_t *trace) {
    ctx_t c; memset(&c, 0, sizeof(c)); memset(res, 0, sizeof(*res));
    const int gn = p->n_rows, qlen = p->qlen;
    if (gn < 3 || qlen < 0) { res->status = ABPOA_HIP_EINVAL; return ABPOA_HIP_EINVAL; }
    int32_t inf;
    c.bits = abpoa_oracle_score_bits(sc, gn, qlen, &inf);
    c.inf = inf; c.pn = c.bits == 16 ? 16 : 8; c.log_n = c.bits == 16 ? 4 : 3;   /* ref: _simd_p16/_simd_p32 :25-29 */
    c.dp_sn = (qlen + c.pn) / c.pn; c.width = c.dp_sn * c.pn;                    /* ref :448 */
    c.P = sc->gap_mode == ABPOA_HIP_LINEAR_GAP ? 1 : (sc->gap_mode == ABPOA_HIP_AFFINE_GAP ? 3 : 5);
    c.local = sc->align_mode == ABPOA_HIP_LOCAL_MODE;
    c.qlen = qlen; c.gn = gn; c.m = sc->m;
    /* gap constants are score_t in the reference (:444, :478, :488-489): truncate the same way */
    c.e1 = W(&c, sc->gap_ext1); c.o1 = W(&c, sc->gap_open1); c.oe1 = W(&c, (long long)sc->gap_open1 + sc->gap_ext1);
    c.e2 = W(&c, sc->gap_ext2); c.o2 = W(&c, sc->gap_open2); c.oe2 = W(&c, (long long)sc->gap_open2 + sc->gap_ext2);
    int w = sc->wb < 0 ? qlen : sc->wb + (int)(sc->wf * qlen);                   /* ref :445, float32 product */
    c.planes = (int32_t *)malloc((size_t)gn * c.P * c.width * sizeof(int32_t));
    c.qp = (int32_t *)malloc((size_t)sc->m * c.width * sizeof(int32_t));
    c.dp_beg = (int *)calloc(gn, sizeof(int)); c.dp_end = (int *)calloc(gn, sizeof(int));
    c.dp_beg_sn = (int *)calloc(gn, sizeof(int)); c.dp_end_sn = (int *)calloc(gn, sizeof(int));
    int *row_max_i = (int *)malloc((size_t)gn * sizeof(int));
    if (!c.planes || !c.qp || !c.dp_beg || !c.dp_end || !c.dp_beg_sn || !c.dp_end_sn || !row_max_i) {
        res->status = ABPOA_HIP_ENOMEM; return ABPOA_HIP_ENOMEM;
    }
    int k, j, row;
    for (row = 0; row < gn; ++row) { row_max_i[row] = -2; c.dp_beg[row] = c.dp_end[row] = c.dp_beg_sn[row] = c.dp_end_sn[row] = -1; } /* -1 = row never computed */
    for (k = 0; k < sc->m; ++k) {                                                /* ref :504-510 */
        int32_t *qk = c.qp + (int64_t)k * c.width;
        qk[0] = 0;
        for (j = 0; j < qlen; ++j) qk[j + 1] = W(&c, sc->mat[k * sc->m + p->query[j]]);
        for (j = qlen + 1; j < c.width; ++j) qk[j] = 0;
    }
    first_row(&c, sc, p, w);
    int best_score = inf, best_i = 0, best_j = 0, best_row_for_zdrop = 0;
    int64_t cells = 0;
    const int need_max = c.local || sc->align_mode == ABPOA_HIP_EXTEND_MODE || sc->wb >= 0;
    for (row = 1; row < gn - 1; ++row) {                                         /* ref :1105 */
        if (p->row_active && !p->row_active[row]) continue;
        dp_row(&c, sc, p, row, w);
        cells += (int64_t)(c.dp_end_sn[row] - c.dp_beg_sn[row] + 1) * c.pn;
        int mx = 0, mi = -1;
        if (need_max) { max_in_row(&c, row, &mx, &mi); row_max_i[row] = mi; }
        if (c.local) {                                                           /* ref :1012-1016, :1108-1110 */
            if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; }
        } else if (sc->align_mode == ABPOA_HIP_EXTEND_MODE) {                     /* ref :1018-1026 */
            if (mx > best_score) { best_score = mx; best_i = row; best_j = mi; best_row_for_zdrop = row; }
            else if (sc->zdrop > 0) {
                int delta_index = p->row_remain[best_row_for_zdrop] - p->row_remain[row];
                if (best_score - mx > sc->zdrop + c.e1 * abs(delta_index - (mi - best_j))) break;
            }
        }
        if (sc->wb >= 0) {                                                       /* ref: simd_abpoa_ada_max_i :1059-1067 */
            int out_i = mi + 1, t;
            for (t = p->out_off[row]; t < p->out_off[row + 1]; ++t) {
                int o = p->out_row[t];
                if (o < 0) continue;
                if (out_i > p->max_pos_right[o]) p->max_pos_right[o] = out_i;
                if (out_i < p->max_pos_left[o]) p->max_pos_left[o] = out_i;
            }
        }
    }
    if (sc->align_mode == ABPOA_HIP_GLOBAL_MODE) {                                /* ref :1028-1041 */
        for (k = p->pred_off[gn - 1]; k < p->pred_off[gn]; ++k) {
            int in_row = p->pred_row[k];
            int end = qlen > c.dp_end[in_row] ? c.dp_end[in_row] : qlen;
            int score = PL(&c, in_row, 0)[end];
            if (score > best_score) { best_score = score; best_i = in_row; best_j = end; }
        }
    }
    res->bits = c.bits; res->best_score = best_score; res->best_row = best_i; res->best_col = best_j;
    res->n_cells = cells;
    int ret = 0;
    if (sc->ret_cigar) ret = backtrack(&c, sc, p, best_i, best_j, res);
    if (trace) {
        trace->bits = c.bits; trace->pn = c.pn; trace->n_planes = c.P; trace->dp_sn = c.dp_sn;
        trace->width = c.width; trace->inf_min = c.inf; trace->n_rows = gn;
        trace->dp_beg = c.dp_beg; trace->dp_end = c.dp_end; trace->dp_beg_sn = c.dp_beg_sn; trace->dp_end_sn = c.dp_end_sn;
        trace->row_max_i = row_max_i; trace->planes = c.planes;
    } else {
        free(c.planes); free(c.dp_beg); free(c.dp_end); free(c.dp_beg_sn); free(c.dp_end_sn); free(row_max_i);
    }
    free(c.qp);
    return ret;
}

void abpoa_oracle_free_trace(abpoa_oracle_trace_t *t) {
    if (!t) return;
    free(t->dp_beg); free(t->dp_end); free(t->dp_beg_sn); free(t->dp_end_sn); free(t->row_max_i); free(t->planes);
    memset(t, 0, sizeof(*t));
}
