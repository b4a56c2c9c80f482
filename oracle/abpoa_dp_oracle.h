/*
 * TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C, scalar restatement of abPOA v1.4.1's banded sequence-to-graph DP
 * (reference: src/simd_abpoa_align.c; every function in abpoa_dp_oracle.c cites the lines it
 * follows).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Parity status: PINNED.  tests/test_oracle_vs_reference.py checks this restatement cell-for-cell
 * (bands, all score planes, best score, cigar) against the compiled reference (oracle/_ref,
 * built by oracle/Makefile from /root/reference) and against the committed golden vectors in
 * tests/golden/ (generated from that reference build by oracle/make_golden.py).
 *
 * It consumes the same flat problem description as the HIP engine (include/abpoa_hip.h) so the
 * GPU parity tests can diff the two on identical inputs.
 */
#ifndef ABPOA_DP_ORACLE_H
#define ABPOA_DP_ORACLE_H

#include "abpoa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Full-width trace, laid out like the reference's abm->s_mem DP area
 * (src/simd_abpoa_align.c:469,480,494): cell (row, plane, col) at
 * planes[((int64_t)row * n_planes + plane) * width + col], width = dp_sn * pn. */
typedef struct abpoa_oracle_trace_t {
    int32_t bits, pn, n_planes, dp_sn, width, inf_min, n_rows;
    int32_t *dp_beg, *dp_end, *dp_beg_sn, *dp_end_sn;   /* [n_rows] */
    int32_t *row_max_i;                                  /* [n_rows], -2 where not computed */
    int32_t *planes;                                     /* sign-extended cells; rows never computed are undefined */
} abpoa_oracle_trace_t;

/* Align one problem.  res->cigar is malloc'ed (caller frees).  If trace != NULL it is filled with
 * malloc'ed arrays (free with abpoa_oracle_free_trace).  Returns 0 or ABPOA_HIP_E*. */
int  abpoa_oracle_align(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p,
                        abpoa_hip_result_t *res, abpoa_oracle_trace_t *trace);
void abpoa_oracle_free_trace(abpoa_oracle_trace_t *t);

/* oracle/dir_model.c: builds the direction plane of abpoa_amd/csrc/dir_plane.h from a trace and walks it (global mode, banded, affine /
 * convex); res->cigar (malloc'ed) must equal abpoa_oracle_align's.  stats[8]: cells, cells in masked-scan vectors, derived-vs-literal F
 * origin mismatches, arithmetic-vs-literal E mismatches, walk steps, steps decided by a literal override, cells of the undecidable class, walk
 * steps whose F origin differs from the reference's comparisons (must be 0). */
int  abpoa_oracle_dir_walk(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, const abpoa_oracle_trace_t *t,
                           int best_i, int best_j, abpoa_hip_result_t *res, int64_t *stats);

int  abpoa_oracle_dir_words(const abpoa_hip_scoring_t *sc, const abpoa_hip_problem_t *p, const abpoa_oracle_trace_t *t, int best_i, int best_j, uint32_t *words);

/* Same arithmetic as abpoa_hip_score_bits (reference src/simd_abpoa_align.c:1672-1683). */
int  abpoa_oracle_score_bits(const abpoa_hip_scoring_t *sc, int n_rows, int qlen, int32_t *inf_min);

#ifdef __cplusplus
}
#endif
#endif
