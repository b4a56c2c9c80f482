/* TEST INFRASTRUCTURE: compile-time proof that include/abpoa_seam.h restates the reference's struct layouts
 * (compiled only where /root/reference exists, by oracle/Makefile). */
#include <stddef.h>
#include "abpoa.h"          /* the reference's own header */
#include "abpoa_seam.h"
#define SAME(a, b) _Static_assert(sizeof(a) == sizeof(b), "sizeof " #a)
#define OFF(a, b, f) _Static_assert(offsetof(a, f) == offsetof(b, f), "offsetof " #f)
SAME(abpoa_res_t, abpoa_seam_res_t); OFF(abpoa_res_t, abpoa_seam_res_t, graph_cigar); OFF(abpoa_res_t, abpoa_seam_res_t, best_score); OFF(abpoa_res_t, abpoa_seam_res_t, n_aln_bases);
SAME(abpoa_para_t, abpoa_seam_para_t); OFF(abpoa_para_t, abpoa_seam_para_t, mat); OFF(abpoa_para_t, abpoa_seam_para_t, wb); OFF(abpoa_para_t, abpoa_seam_para_t, wf);
OFF(abpoa_para_t, abpoa_seam_para_t, zdrop); OFF(abpoa_para_t, abpoa_seam_para_t, align_mode); OFF(abpoa_para_t, abpoa_seam_para_t, gap_mode); OFF(abpoa_para_t, abpoa_seam_para_t, min_freq); OFF(abpoa_para_t, abpoa_seam_para_t, incr_fn);
SAME(abpoa_node_t, abpoa_seam_node_t); OFF(abpoa_node_t, abpoa_seam_node_t, in_id); OFF(abpoa_node_t, abpoa_seam_node_t, out_id); OFF(abpoa_node_t, abpoa_seam_node_t, base); OFF(abpoa_node_t, abpoa_seam_node_t, aligned_node_id);
SAME(abpoa_graph_t, abpoa_seam_graph_t); OFF(abpoa_graph_t, abpoa_seam_graph_t, index_to_node_id); OFF(abpoa_graph_t, abpoa_seam_graph_t, node_id_to_max_remain); OFF(abpoa_graph_t, abpoa_seam_graph_t, node_id_to_max_pos_right);
SAME(abpoa_simd_matrix_t, abpoa_seam_matrix_t); OFF(abpoa_simd_matrix_t, abpoa_seam_matrix_t, rang_m);
SAME(abpoa_t, abpoa_seam_t); OFF(abpoa_t, abpoa_seam_t, abm);
int main(void) { return 0; }
