/*
 * abpoa_seam.h -- the narrowest drop-in seam: the four symbols of abPOA v1.4.1's
 * src/simd_abpoa_align.h:11-14, re-implemented on the MI355X engine.
 *
 * Linking the stock reference objects (everything except src/simd_abpoa_align.o) against
 * libabpoa_hip.so gives the unmodified abpoa binary / library / pyabpoa a GPU-backed DP:
 *
 *   simd_abpoa_align_sequence_to_graph      (reference src/simd_abpoa_align.c:1714)
 *   simd_abpoa_align_sequence_to_subgraph   (reference src/simd_abpoa_align.c:1645)
 *   abpoa_init_simd_matrix                  (reference src/simd_abpoa_align.c:1159)
 *   abpoa_free_simd_matrix                  (reference src/simd_abpoa_align.c:1166)
 *
 * The structs below restate the MEMORY LAYOUT of the reference's public types (src/abpoa.h:53-135)
 * under seam-local names, so that this header can coexist with the reference's abpoa.h in one program;
 * a reference-side caller simply passes its own abpoa_t* / abpoa_para_t* / abpoa_res_t*.
 * Only the fields the DP path reads or writes are named individually in the comments.
 */
#ifndef ABPOA_SEAM_H
#define ABPOA_SEAM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {                       /* abpoa_res_t, src/abpoa.h:53-60 */
    int n_cigar, m_cigar; uint64_t *graph_cigar;      /* callee (re)allocates with libc malloc; caller frees */
    int node_s, node_e, query_s, query_e;
    int n_aln_bases, n_matched_bases;                 /* accumulated, never reset by the callee (:132) */
    int32_t best_score;
} abpoa_seam_res_t;

typedef struct {                       /* abpoa_para_t, src/abpoa.h:62-81 */
    int m; int *mat; char *mat_fn;
    int use_score_matrix;
    int match, max_mat, mismatch, min_mis, gap_open1, gap_open2, gap_ext1, gap_ext2; int inf_min;
    int k, w, min_w;
    int wb; float wf;
    int zdrop, end_bonus;
    uint8_t ret_cigar:1, rev_cigar:1, out_msa:1, out_cons:1, out_gfa:1, out_fq:1, use_read_ids:1, amb_strand:1;
    uint8_t use_qv:1, disable_seeding:1, progressive_poa:1;
    char *incr_fn, *out_pog;
    int align_mode, gap_mode, max_n_cons;
    double min_freq;
    int verbose;
} abpoa_seam_para_t;

typedef struct {                       /* abpoa_node_t, src/abpoa.h:83-94 */
    int node_id;
    int in_edge_n, in_edge_m, *in_id;
    int out_edge_n, out_edge_m, *out_id; int *out_weight;
    int *read_weight, n_read, m_read;
    uint64_t **read_ids; int read_ids_n;
    int aligned_node_n, aligned_node_m, *aligned_node_id;
    uint8_t base;
} abpoa_seam_node_t;

typedef struct {                       /* abpoa_graph_t, src/abpoa.h:96-101 */
    abpoa_seam_node_t *node; int node_n, node_m, index_rank_m;
    int *index_to_node_id;
    int *node_id_to_index, *node_id_to_max_pos_left, *node_id_to_max_pos_right, *node_id_to_max_remain, *node_id_to_msa_rank;
    uint8_t is_topological_sorted:1, is_called_cons:1, is_set_msa_rank:1;
} abpoa_seam_graph_t;

typedef struct {                       /* abpoa_simd_matrix_t, src/abpoa.h:125-128: owned by the seam, opaque to callers */
    void *s_mem; uint64_t s_msize;
    int *dp_beg, *dp_end, *dp_beg_sn, *dp_end_sn, rang_m;
} abpoa_seam_matrix_t;

typedef struct {                       /* abpoa_t, src/abpoa.h:130-135 */
    abpoa_seam_graph_t *abg;
    void *abs;
    abpoa_seam_matrix_t *abm;
    void *abc;
} abpoa_seam_t;

/* Same contracts as the reference: always return 0; fatal conditions print to stderr and exit(1)
 * (src/utils.c:91-116) -- including "no usable GPU", since there is no CPU fallback. */
int simd_abpoa_align_sequence_to_graph(abpoa_seam_t *ab, abpoa_seam_para_t *abpt, uint8_t *query, int qlen, abpoa_seam_res_t *res);
int simd_abpoa_align_sequence_to_subgraph(abpoa_seam_t *ab, abpoa_seam_para_t *abpt, int beg_node_id, int end_node_id,
                                          uint8_t *query, int qlen, abpoa_seam_res_t *res);
abpoa_seam_matrix_t *abpoa_init_simd_matrix(void);
void abpoa_free_simd_matrix(abpoa_seam_matrix_t *abm);

#ifdef __cplusplus
}
#endif
#endif
