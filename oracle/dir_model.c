/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 *
 * CPU model of the direction plane (abpoa_amd/csrc/dir_plane.h): from a full oracle trace (every H / E / F value of an alignment) it
 * builds the per-cell words exactly as the HIP row loops define them -- first-arg-max predecessor indices with the score ring's
 * out-of-band-reads-inf rule, the saturated differences uE / dF, the literal F-origin override in the vectors where the reference's
 * masked F scan (src/simd_abpoa_align.c:665-699) applies -- and walks them with the decision order of the reference backtrack
 * (:109-429).  The cigar it produces must equal the oracle's own, value-comparing backtrack (abpoa_dp_oracle.c, pinned against the
 * compiled reference); tests/test_dir_model.py checks that on every golden and on seeded read-sets.  This is the executable
 * specification the GPU walker (abpoa_amd/csrc/backtrack_dir.h) follows; it also counts how often the cheap rules the row loops
 * use (derived F origin, arithmetic uE) disagree with the literal comparisons on cells that hold real scores.
 */
#include <limits.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include "abpoa_dp_oracle.h"
#include "../abpoa_amd/csrc/dir_plane.h"

#define OP_M 0x1
#define OP_E1 0x2
#define OP_E2 0x4
#define OP_E 0x6
#define OP_F1 0x8
#define OP_F2 0x10
#define OP_F 0x18
#define OP_ALL 0x1f

typedef struct { uint64_t *a; int n, m; } cig_t;
static void push_cigar(cig_t *cg, int op, int len, int32_t node_id, int32_t query_id) {      /* ref: abpoa_align.h:54-73 */
    uint64_t l = (uint64_t)(int64_t)len;
    if (cg->n == 0 || op != ABPOA_HIP_CINS || op != (int)(cg->a[cg->n - 1] & 0xf)) {
        if (cg->n == cg->m) { cg->m = cg->m ? cg->m << 1 : 4; cg->a = (uint64_t *)realloc(cg->a, (size_t)cg->m * sizeof(uint64_t)); }
        uint64_t n_id = (uint64_t)(int64_t)node_id, q_id = (uint64_t)(int64_t)query_id;
        if (op == ABPOA_HIP_CMATCH) cg->a[cg->n++] = n_id << 34 | q_id << 4 | (uint64_t)op;
        else if (op == ABPOA_HIP_CINS) cg->a[cg->n++] = q_id << 34 | l << 4 | (uint64_t)op;
        else cg->a[cg->n++] = n_id << 34 | l << 4 | (uint64_t)op;
    } else cg->a[cg->n - 1] += l << 4;
}

typedef struct {
    const abpoa_hip_scoring_t *sc; const abpoa_hip_problem_t *p; const abpoa_oracle_trace_t *t;
    int convex, pn, inf, e1, oe1, o1, e2, oe2, o2;
    int pE1, pE2, pF1, pF2;
    uint32_t **rows;       /* words of row i, column dp_beg[i] + x */
    uint32_t *words_out;   /* NULL, or [n_rows * width] */
    int32_t *hv_row;       /* H before the F terms are merged in (max of the match and E terms) of the row being built, by column - dp_beg */
    int64_t *stats;
} model_t;

static inline int wrapw(const model_t *m, long long x) { return m->t->bits == 16 ? (int)(int16_t)(uint16_t)(uint64_t)x : (int)(int32_t)(uint32_t)(uint64_t)x; }
static inline const int32_t *PLN(const model_t *m, int row, int plane) { return m->t->planes + ((int64_t)row * m->t->n_planes + plane) * m->t->width; }
static inline int in_band(const model_t *m, int row, int col) { return m->t->dp_beg_sn[row] >= 0 && col >= m->t->dp_beg[row] && col <= m->t->dp_end[row]; }

/* stats: 0 cells, 1 cells in masked-scan vectors, 2 fast cells whose derived F origin differs from the literal comparison (F real),
 *        3 cells whose arithmetic uE disagrees with the literal comparisons (E real), 4 walk steps, 5 walk steps that used a literal override,
 *        6 cells of the class the word cannot decide (see dir_plane.h), 7 walk steps on which the word's F origin differs from the reference's comparisons */
static uint32_t cell_word(const model_t *m, int i, int j, int max_pre_end_sn) {
    const abpoa_hip_problem_t *p = m->p; const abpoa_oracle_trace_t *t = m->t;
    const int pn = m->pn, inf = m->inf;
    const int np = p->pred_off[i + 1] - p->pred_off[i]; const int *preds = p->pred_row + p->pred_off[i];
    const int32_t *H = PLN(m, i, 0), *E1 = PLN(m, i, m->pE1), *F1 = PLN(m, i, m->pF1);
    const int32_t *F2 = m->convex ? PLN(m, i, m->pF2) : NULL;
    int Mv = inf, E1v = inf, E2v = inf, kf = 1, kE1 = 1, kE2 = 1, k;
    for (k = 0; k < np; ++k) {      /* the row loops' gather: the first predecessor unmasked (its score-ring row reads "inf" outside its band), the others range-checked */
        const int pr = preds[k], pb = t->dp_beg_sn[pr] * pn, Wp = (t->dp_end_sn[pr] - t->dp_beg_sn[pr] + 1) * pn, x = j - pb;
        const int hm1 = (x - 1 >= 0 && x - 1 < Wp) ? PLN(m, pr, 0)[j - 1] : inf;
        const int ev1 = (x >= 0 && x < Wp) ? PLN(m, pr, m->pE1)[j] : inf;
        const int ev2 = (m->convex && x >= 0 && x < Wp) ? PLN(m, pr, m->pE2)[j] : inf;
        if (k == 0) { Mv = hm1; E1v = ev1; E2v = ev2; }
        else {
            const int inH = x >= 0 && x < Wp + pn, inE = x >= 0 && x < Wp;
            if (inH && hm1 > Mv) { kf = k + 1; Mv = hm1; }
            if (inE && ev1 > E1v) { kE1 = k + 1; E1v = ev1; }
            if (inE && ev2 > E2v) { kE2 = k + 1; E2v = ev2; }
        }
    }
    const int q = (j >= 1 && j <= p->qlen) ? m->sc->mat[m->sc->m * p->row_base[i] + p->query[j - 1]] : 0;      /* (columns past the query score 0, reference query profile :504-510) */
    const int Hout = H[j];
    /* what the reference's F recurrence opens from: the match term alone in affine mode (:870 uses H before the E merge), max(match, E1, E2) in convex mode (:987-990) */
    { int hvv = wrapw(m, (long long)Mv + q); if (m->convex && E1v > hvv) hvv = E1v; if (m->convex && E2v > hvv) hvv = E2v; m->hv_row[j - t->dp_beg[i]] = hvv; }
    const int kM = (wrapw(m, (long long)Mv + q) == Hout) ? kf : 0;
    /* uE: max(o - (H - Ein), 0) computed as the row loops do: en - (H - oe) with en = max(Ein - e, H - oe) */
    const int t2a = wrapw(m, (long long)Hout - m->oe1), en1 = wrapw(m, (long long)E1v - m->e1) > t2a ? wrapw(m, (long long)E1v - m->e1) : t2a;
    int uE1 = en1 - t2a; if (uE1 < 0) uE1 = 0; if (uE1 > m->o1) uE1 = m->o1;
    int uE2 = 0;
    if (m->convex) { const int t2b = wrapw(m, (long long)Hout - m->oe2), en2 = wrapw(m, (long long)E2v - m->e2) > t2b ? wrapw(m, (long long)E2v - m->e2) : t2b; uE2 = en2 - t2b; if (uE2 < 0) uE2 = 0; if (uE2 > m->o2) uE2 = m->o2; }
    {   /* literal cross-check on cells whose incoming E is a real score */
        const int real = E1v > inf + 64 * (m->e1 > m->e2 ? m->e1 : m->e2) && Hout > inf + 4096;
        if (real) {
            const int hE = Hout == E1v, oE = E1[j] == t2a;
            if ((uE1 == m->o1) != hE) m->stats[3]++;
            else if (!m->convex) { if (Hout == (Hout > E1v ? Hout : E1v) && (uE1 == 0) != oE && E1[j] != inf) m->stats[3]++; }
            else if ((uE1 == 0) != oE) m->stats[3]++;
        }
    }
    long long d1 = (long long)Hout - F1[j]; if (d1 < 0) d1 = 0; const int capA = m->convex ? DIRC_CAP1 : DIRA_CAP1; if (d1 > capA) d1 = capA;
    long long d2 = 0; if (m->convex) { d2 = (long long)Hout - F2[j]; if (d2 < 0) d2 = 0; if (d2 > DIRC_CAP2) d2 = DIRC_CAP2; }
    /* literal F origin: only in the vectors of the masked scan */
    int l1 = DIR_LIT_NONE, l2 = DIR_LIT_NONE;
    const int v = j / pn, stored_jm1 = j - 1 >= t->dp_beg[i];
    if (stored_jm1) {
        const int lo1 = wrapw(m, (long long)H[j - 1] - m->oe1) == F1[j] ? DIR_LIT_OPEN : (wrapw(m, (long long)F1[j - 1] - m->e1) == F1[j] ? DIR_LIT_EXT : DIR_LIT_NEITHER);
        int lo2 = DIR_LIT_NONE;
        if (m->convex) lo2 = wrapw(m, (long long)H[j - 1] - m->oe2) == F2[j] ? DIR_LIT_OPEN : (wrapw(m, (long long)F2[j - 1] - m->e2) == F2[j] ? DIR_LIT_EXT : DIR_LIT_NEITHER);
        if (v > max_pre_end_sn) { l1 = lo1; l2 = lo2; m->stats[1]++; }
        else {      /* closed-form vectors: the rule dir_f_origin() on column j-1 must agree with the literal comparisons wherever F is a real score; the
                       one undecidable class (H[j-1] an F term, dF > o, literal "neither") is counted apart -- the walk must never consult it */
            const int32_t *E1m = PLN(m, i, m->pE1);
            (void)E1m;
            const int hv1 = H[j - 1] == m->hv_row[j - 1 - t->dp_beg[i]];
            if (F1[j] > inf + 4096) { const long long dd = (long long)H[j - 1] - F1[j - 1]; const int der = dir_f_origin(hv1, dd > 64 ? 64 : (int)dd, m->o1);
                                      if (!hv1 && dd > m->o1 && lo1 == DIR_LIT_NEITHER) m->stats[6]++; else if (der != lo1) { if (getenv("DIR_MODEL_DEBUG") && m->stats[2] < 5) fprintf(stderr, "row %d col %d (v %d, max_pre_end_sn %d, beg %d): H[j-1] %d hv[j-1] %d F1[j-1] %d F1[j] %d H[j] %d derived %d literal %d\n", i, j, v, max_pre_end_sn, t->dp_beg[i], H[j-1], m->hv_row[j - 1 - t->dp_beg[i]], F1[j-1], F1[j], H[j], der, lo1); m->stats[2]++; } }
            if (m->convex && F2[j] > inf + 4096) { const long long dd = (long long)H[j - 1] - F2[j - 1]; const int der = dir_f_origin(hv1, dd > 64 ? 64 : (int)dd, m->o2);
                                                   if (!hv1 && dd > m->o2 && lo2 == DIR_LIT_NEITHER) m->stats[6]++; else if (der != lo2) m->stats[2]++; }
        }
    }
    m->stats[0]++;
    if (!m->convex) return (uint32_t)kM << DIRA_KM_SH | (uint32_t)kE1 << DIRA_KE1_SH | (uint32_t)uE1 << DIRA_UE1_SH | (uint32_t)d1 << DIRA_DF1_SH | (uint32_t)l1 << DIRA_LF1_SH;
    return (uint32_t)kM << DIRC_KM_SH | (uint32_t)kE1 << DIRC_KE1_SH | (uint32_t)kE2 << DIRC_KE2_SH | (uint32_t)uE1 << DIRC_UE1_SH | (uint32_t)uE2 << DIRC_UE2_SH |
           (uint32_t)d1 << DIRC_DF1_SH | (uint32_t)d2 << DIRC_DF2_SH | (uint32_t)l1 << DIRC_LF1_SH | (uint32_t)l2 << DIRC_LF2_SH;
}

typedef struct { int kM, kE[2], uE[2], dF[2], lF[2]; } dw_t;
static dw_t decode(const model_t *m, uint32_t w) {
    dw_t d; memset(&d, 0, sizeof(d));
    if (!m->convex) { d.kM = (w >> DIRA_KM_SH) & 15; d.kE[0] = (w >> DIRA_KE1_SH) & 15; d.uE[0] = (w >> DIRA_UE1_SH) & 7; d.dF[0] = (w >> DIRA_DF1_SH) & 7; d.lF[0] = (w >> DIRA_LF1_SH) & 3; }
    else { d.kM = (w >> DIRC_KM_SH) & 15; d.kE[0] = (w >> DIRC_KE1_SH) & 15; d.kE[1] = (w >> DIRC_KE2_SH) & 15; d.uE[0] = (w >> DIRC_UE1_SH) & 7; d.uE[1] = (w >> DIRC_UE2_SH) & 31;
           d.dF[0] = (w >> DIRC_DF1_SH) & 7; d.dF[1] = (w >> DIRC_DF2_SH) & 31; d.lF[0] = (w >> DIRC_LF1_SH) & 3; d.lF[1] = (w >> DIRC_LF2_SH) & 3; }
    return d;
}

/* Global mode, banded, affine / convex.  Fills res (cigar malloc'ed; fields as the oracle's backtrack) and stats[8].
 * Returns 0, ABPOA_HIP_EINVAL when the direction plane does not apply, ABPOA_HIP_EBACKTRACK on a dead end. */
static uint32_t *g_words_out = NULL;
int abpoa_oracle_dir_walk(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, const abpoa_oracle_trace_t *t,
                          int best_i, int best_j, abpoa_hip_result_t *res, int64_t *stats);
/* the words of every cell (row-major, t->width columns per row, 0 outside the bands), as the row loops must write them -- except that the model takes
 * the literal F-origin override only in the masked-scan vectors, while the exact row bodies of the kernels write it everywhere (compare with the lF fields masked) */
int abpoa_oracle_dir_words(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, const abpoa_oracle_trace_t *t, int best_i, int best_j, uint32_t *words) {
    abpoa_hip_result_t r; int64_t st[10];
    memset(words, 0, (size_t)p->n_rows * t->width * sizeof(uint32_t));
    g_words_out = words;
    const int rc = abpoa_oracle_dir_walk(sc, p, t, best_i, best_j, &r, st);
    g_words_out = NULL;
    if (rc == 0) free(r.cigar);
    return rc;
}
int abpoa_oracle_dir_walk(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, const abpoa_oracle_trace_t *t,
                          int best_i, int best_j, abpoa_hip_result_t *res, int64_t *stats) {
    model_t M; memset(&M, 0, sizeof(M)); M.words_out = g_words_out; memset(stats, 0, 10 * sizeof(int64_t));
    if (sc->align_mode != ABPOA_HIP_GLOBAL_MODE || sc->wb < 0 || sc->gap_mode == ABPOA_HIP_LINEAR_GAP || !dir_plane_usable(sc->gap_mode, sc->gap_open1, sc->gap_ext1, sc->gap_open2, sc->gap_ext2)) return ABPOA_HIP_EINVAL;
    const int gn = p->n_rows, qlen = p->qlen; int i, j, k;
    for (i = 0; i < gn; ++i) if (p->pred_off[i + 1] - p->pred_off[i] > DIR_K_MAX) return ABPOA_HIP_EINVAL;
    M.sc = sc; M.p = p; M.t = t; M.stats = stats; M.convex = sc->gap_mode == ABPOA_HIP_CONVEX_GAP; M.pn = t->pn; M.inf = t->inf_min;
    M.e1 = wrapw(&M, sc->gap_ext1); M.o1 = wrapw(&M, sc->gap_open1); M.oe1 = wrapw(&M, (long long)sc->gap_open1 + sc->gap_ext1);
    M.e2 = wrapw(&M, sc->gap_ext2); M.o2 = wrapw(&M, sc->gap_open2); M.oe2 = wrapw(&M, (long long)sc->gap_open2 + sc->gap_ext2);
    M.pE1 = 1; M.pE2 = 2; M.pF1 = M.convex ? 3 : 2; M.pF2 = 4;
    M.rows = (uint32_t **)calloc((size_t)gn, sizeof(uint32_t *));
    for (i = 1; i < gn - 1; ++i) {
        if (t->dp_beg_sn[i] < 0) continue;
        int max_pre_end_sn = -1;
        for (k = p->pred_off[i]; k < p->pred_off[i + 1]; ++k) if (t->dp_end_sn[p->pred_row[k]] > max_pre_end_sn) max_pre_end_sn = t->dp_end_sn[p->pred_row[k]];
        const int W = t->dp_end[i] - t->dp_beg[i] + 1;
        M.rows[i] = (uint32_t *)malloc((size_t)W * sizeof(uint32_t));
        M.hv_row = (int32_t *)realloc(M.hv_row, (size_t)W * sizeof(int32_t));
        for (j = t->dp_beg[i]; j <= t->dp_end[i]; ++j) M.rows[i][j - t->dp_beg[i]] = cell_word(&M, i, j, max_pre_end_sn);
    }
    if (M.words_out) {      /* (abpoa_oracle_dir_words: the caller wants the words themselves, full-width rows of t->width columns, 0 outside the bands) */
        for (i = 1; i < gn - 1; ++i) if (M.rows[i]) for (j = t->dp_beg[i]; j <= t->dp_end[i]; ++j) M.words_out[(int64_t)i * t->width + j] = M.rows[i][j - t->dp_beg[i]];
    }
    /* ---- the walk: reference order (:109-429), decisions from the words only */
    cig_t cg = {0, 0, 0};
    int start_i = best_i, start_j = best_j, cur_op = OP_ALL, indel_first = 1, n_aln = 0, n_match = 0, ret = 0;
    i = best_i; j = best_j;
    if (best_j < qlen) push_cigar(&cg, ABPOA_HIP_CINS, qlen - j, -1, qlen - 1);
    while (i > 0 && j > 0) {
        if (!M.rows[i] || !in_band(&M, i, j)) { ret = ABPOA_HIP_EBACKTRACK; break; }
        const dw_t d = decode(&M, M.rows[i][j - t->dp_beg[i]]);
        const int np = p->pred_off[i + 1] - p->pred_off[i]; const int *preds = p->pred_row + p->pred_off[i];
        const int id = p->row_node_id[i], is_match = p->row_base[i] == p->query[j - 1];
        int hit = 0, x;
        start_i = i; start_j = j; stats[4]++;
#define TRY_MATCH(set_indel) do { if (d.kM >= 1 && d.kM <= np && in_band(&M, preds[d.kM - 1], j - 1)) {                \
            cur_op = OP_ALL; hit = 1; push_cigar(&cg, ABPOA_HIP_CMATCH, 1, id, j - 1);                                 \
            i = preds[d.kM - 1]; --j; ++n_aln; n_match += is_match; if (set_indel) indel_first = 0; } } while (0)
        if ((cur_op & OP_M) && indel_first == 0) TRY_MATCH(0);
        if (!hit && (cur_op & OP_E)) {
            const int viaM = cur_op & OP_M; int kk[2] = {INT_MAX, INT_MAX};
            for (x = 0; x < (M.convex ? 2 : 1); ++x) {
                if (!(cur_op & (x == 0 ? OP_E1 : OP_E2))) continue;
                const int ox = x == 0 ? M.o1 : M.o2;
                if (d.kE[x] < 1 || d.kE[x] > np || !in_band(&M, preds[d.kE[x] - 1], j)) continue;
                if (viaM && d.uE[x] != ox) continue;       /* H == pre_E;  without M: E == pre_E - e holds for the arg-max predecessor by construction */
                kk[x] = d.kE[x];
            }
            if (kk[0] != INT_MAX || kk[1] != INT_MAX) {
                const int use1 = kk[0] <= kk[1], ks = use1 ? kk[0] : kk[1], pr = preds[ks - 1];
                int opened = 0;
                if (pr > 0) { if (!M.rows[pr] || !in_band(&M, pr, j)) { ret = ABPOA_HIP_EBACKTRACK; break; } const dw_t dp = decode(&M, M.rows[pr][j - t->dp_beg[pr]]); opened = dp.uE[use1 ? 0 : 1] == 0; }
                cur_op = opened ? (OP_M | OP_F) : (use1 ? OP_E1 : OP_E2);
                hit = 1; push_cigar(&cg, ABPOA_HIP_CDEL, 1, id, j - 1); i = pr;
            }
        }
        if (!hit && (cur_op & OP_F)) {
            for (x = 0; x < (M.convex ? 2 : 1) && !hit; ++x) {
                const int bit = x == 0 ? OP_F1 : OP_F2, ox = x == 0 ? M.o1 : M.o2;
                if (!(cur_op & bit)) continue;
                if ((cur_op & OP_M) && d.dF[x] != 0) continue;              /* H == F */
                if (j - 1 < t->dp_beg[i]) continue;                        /* column j-1 not stored */
                int lit = d.lF[x];
                if (lit == DIR_LIT_NONE) {
                    const dw_t dl = decode(&M, M.rows[i][j - 1 - t->dp_beg[i]]);
                    const int h_is_hv = dl.kM != 0 || (M.convex && (dl.uE[0] == M.o1 || dl.uE[1] == M.o2));
                    lit = dir_f_origin(h_is_hv, dl.dF[x], ox);
                    if (!h_is_hv && dl.dF[x] > ox) stats[8]++;
                    {   /* what the reference's comparisons say here (the model has the scores): must agree on every step actually taken */
                        const int32_t *Hh = PLN(&M, i, 0), *Fx = PLN(&M, i, x == 0 ? M.pF1 : M.pF2); const int oex = x == 0 ? M.oe1 : M.oe2, ex = x == 0 ? M.e1 : M.e2;
                        const int lref = wrapw(&M, (long long)Hh[j - 1] - oex) == Fx[j] ? DIR_LIT_OPEN : (wrapw(&M, (long long)Fx[j - 1] - ex) == Fx[j] ? DIR_LIT_EXT : DIR_LIT_NEITHER);
                        if (lref != lit) stats[7]++;
                    }
                } else stats[5]++;
                if (lit == DIR_LIT_OPEN) { cur_op = OP_M | OP_E; hit = 1; }
                else if (lit == DIR_LIT_EXT) { cur_op = bit; hit = 1; }
            }
            if (hit) { push_cigar(&cg, ABPOA_HIP_CINS, 1, id, j - 1); --j; ++n_aln; }
        }
        if (!hit && (cur_op & OP_M) && indel_first == 1) TRY_MATCH(1);
#undef TRY_MATCH
        if (!hit) { ret = ABPOA_HIP_EBACKTRACK; break; }
    }
    for (i = 0; i < gn; ++i) free(M.rows[i]);
    free(M.rows); free(M.hv_row);
    memset(res, 0, sizeof(*res));
    if (ret) { free(cg.a); res->status = ret; return ret; }
    if (j > 0) push_cigar(&cg, ABPOA_HIP_CINS, j, -1, j - 1);
    if (!sc->rev_cigar) for (k = 0; k < cg.n >> 1; ++k) { uint64_t tt = cg.a[k]; cg.a[k] = cg.a[cg.n - 1 - k]; cg.a[cg.n - 1 - k] = tt; }
    res->cigar = cg.a; res->n_cigar = cg.n; res->best_row = best_i; res->best_col = best_j;
    res->node_e = p->row_node_id[best_i]; res->query_e = best_j - 1; res->node_s = p->row_node_id[start_i]; res->query_s = start_j - 1;
    res->n_aln_bases = n_aln; res->n_matched_bases = n_match;
    return 0;
}
