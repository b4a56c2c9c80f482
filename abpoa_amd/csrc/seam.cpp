// Seam API (include/abpoa_seam.h): the four symbols of the reference's src/simd_abpoa_align.h on top of the flat
// batch API.  The reference's graph (abpoa_graph_t) is flattened into DP rows exactly as its own DP reads it:
// index_map reachability (src/simd_abpoa_align.c:1650-1660), predecessor lists filtered by it (:519-530), and the
// band state node_id_to_max_pos_left/right is gathered before and scattered back after the call (the DP mutates it,
// :556-561, :1059-1067 -- also for successors outside a sub-graph window).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "engine_options.h"
#include "../../include/abpoa_hip.h"
#include "../../include/abpoa_seam.h"

static void seam_fatal(const char *fn, const char *msg) {          // reference err_fatal: message + exit(1), src/utils.c:91-116
    fprintf(stderr, "[%s] %s\n", fn, msg); exit(EXIT_FAILURE);
}

extern "C" {

abpoa_seam_matrix_t *abpoa_init_simd_matrix(void) {
    abpoa_seam_matrix_t *abm = (abpoa_seam_matrix_t *)calloc(1, sizeof(abpoa_seam_matrix_t));
    if (!abm) seam_fatal(__func__, "out of memory");
    return abm;           // the score planes live in HBM; nothing to allocate on the host
}

void abpoa_free_simd_matrix(abpoa_seam_matrix_t *abm) {
    if (!abm) return;
    free(abm->dp_beg); free(abm->dp_end); free(abm->dp_beg_sn); free(abm->dp_end_sn); free(abm);
}

int simd_abpoa_align_sequence_to_subgraph(abpoa_seam_t *ab, abpoa_seam_para_t *abpt, int beg_node_id, int end_node_id,
                                          uint8_t *query, int qlen, abpoa_seam_res_t *res) {
    abpoa_hip::refresh_options();
    abpoa_seam_graph_t *g = ab->abg;
    if (beg_node_id < 0 || beg_node_id >= g->node_n || end_node_id < 0 || end_node_id >= g->node_n) seam_fatal(__func__, "Wrong node id");
    const int beg_index = g->node_id_to_index[beg_node_id], end_index = g->node_id_to_index[end_node_id];
    const int gn = end_index - beg_index + 1;
    const bool banded = abpt->wb >= 0;
    if (gn < 3) { res->best_score = 0; return 0; }
    // index_map, reference :1650-1660
    std::vector<uint8_t> index_map(g->node_n, 0);
    index_map[beg_index] = index_map[end_index] = 1;
    for (int i = beg_index; i < end_index - 1; ++i) {
        if (!index_map[i]) continue;
        const abpoa_seam_node_t &nd = g->node[g->index_to_node_id[i]];
        for (int j = 0; j < nd.out_edge_n; ++j) index_map[g->node_id_to_index[nd.out_id[j]]] = 1;
    }
    std::vector<uint8_t> base(gn), active(gn);
    std::vector<int32_t> nid(gn), remain(gn, 0), left(gn, 0), right(gn, 0), poff(gn + 1), ooff(gn + 1), pred, out;
    for (int r = 0; r < gn; ++r) {
        const int id = g->index_to_node_id[beg_index + r]; const abpoa_seam_node_t &nd = g->node[id];
        base[r] = nd.base; nid[r] = id; active[r] = index_map[beg_index + r];
        if ((banded || abpt->zdrop > 0) && g->node_id_to_max_remain) remain[r] = g->node_id_to_max_remain[id];
        if (banded) { left[r] = g->node_id_to_max_pos_left[id]; right[r] = g->node_id_to_max_pos_right[id]; }
        poff[r] = (int)pred.size(); ooff[r] = (int)out.size();
        if (r > 0) for (int j = 0; j < nd.in_edge_n; ++j) {           // reference :519-530
            const int pi = g->node_id_to_index[nd.in_id[j]];
            if (index_map[pi]) pred.push_back(pi - beg_index);
        }
        for (int j = 0; j < nd.out_edge_n; ++j) {
            const int oi = g->node_id_to_index[nd.out_id[j]];
            out.push_back((oi >= beg_index && oi <= end_index) ? oi - beg_index : -1);
        }
    }
    poff[gn] = (int)pred.size(); ooff[gn] = (int)out.size();
    if (pred.empty()) pred.push_back(0);
    if (out.empty()) out.push_back(0);
    active[gn - 1] = 1;
    abpoa_hip_scoring_t sc;
    sc.m = abpt->m; sc.mat = abpt->mat; sc.max_mat = abpt->max_mat; sc.min_mis = abpt->min_mis;
    sc.gap_open1 = abpt->gap_open1; sc.gap_ext1 = abpt->gap_ext1; sc.gap_open2 = abpt->gap_open2; sc.gap_ext2 = abpt->gap_ext2;
    sc.align_mode = abpt->align_mode; sc.gap_mode = abpt->gap_mode; sc.wb = abpt->wb; sc.wf = abpt->wf; sc.zdrop = abpt->zdrop;
    sc.ret_cigar = abpt->ret_cigar; sc.rev_cigar = abpt->rev_cigar;
    abpoa_hip_problem_t p;
    p.n_rows = gn; p.qlen = qlen; p.query = query; p.row_base = base.data(); p.row_node_id = nid.data(); p.row_remain = remain.data();
    p.row_active = active.data(); p.pred_off = poff.data(); p.pred_row = pred.data(); p.out_off = ooff.data(); p.out_row = out.data();
    p.max_pos_left = left.data(); p.max_pos_right = right.data();
    abpoa_hip_result_t r;
    // ask for the trace only when a sub-graph window has successors outside it: their band state is updated from row_max_i
    bool outside = false;
    for (size_t t = 0; t < out.size() && !outside; ++t) outside = out[t] < 0;
    const int rc = abpoa_hip_align_batch(&sc, 1, &p, &r, (banded && outside) ? ABPOA_HIP_FLAG_TRACE : 0);
    if (rc != ABPOA_HIP_OK) seam_fatal(__func__, abpoa_hip_last_error());
    if (r.status == ABPOA_HIP_EBACKTRACK) seam_fatal(__func__, "Error in backtrack.");     // reference :171 / :275 / :419
    if (r.status != 0) seam_fatal(__func__, "alignment failed on the device");
    if (banded) {
        for (int rr = 0; rr < gn; ++rr) { const int id = nid[rr]; g->node_id_to_max_pos_left[id] = left[rr]; g->node_id_to_max_pos_right[id] = right[rr]; }
        if (outside) {       // reference :557-561 and :1059-1067 also touch out-nodes beyond the window
            const abpoa_seam_node_t &bn = g->node[beg_node_id];
            for (int j = 0; j < bn.out_edge_n; ++j) {
                const int oi = g->node_id_to_index[bn.out_id[j]];
                if ((oi < beg_index || oi > end_index) && index_map[oi]) g->node_id_to_max_pos_left[bn.out_id[j]] = g->node_id_to_max_pos_right[bn.out_id[j]] = 1;
            }
            for (int rr = 1; rr < gn - 1; ++rr) {
                if (!active[rr] || !r.trace || r.trace->dp_beg_sn[rr] < 0) continue;
                const int out_i = r.trace->row_max_i[rr] + 1; const abpoa_seam_node_t &nd = g->node[nid[rr]];
                for (int j = 0; j < nd.out_edge_n; ++j) {
                    const int oid = nd.out_id[j], oi = g->node_id_to_index[oid];
                    if (oi >= beg_index && oi <= end_index) continue;
                    if (out_i > g->node_id_to_max_pos_right[oid]) g->node_id_to_max_pos_right[oid] = out_i;
                    if (out_i < g->node_id_to_max_pos_left[oid]) g->node_id_to_max_pos_left[oid] = out_i;
                }
            }
        }
    }
    res->best_score = r.best_score;
    if (abpt->ret_cigar) {
        res->graph_cigar = r.cigar; r.cigar = nullptr;               // libc malloc'ed by the engine: the caller frees it
        res->n_cigar = r.n_cigar; res->m_cigar = r.n_cigar;
        res->node_s = r.node_s; res->node_e = r.node_e; res->query_s = r.query_s; res->query_e = r.query_e;
        res->n_aln_bases += r.n_aln_bases; res->n_matched_bases += r.n_matched_bases;     // accumulated like the reference (:132)
    }
    abpoa_hip_free_result(&r);
    return 0;
}

int simd_abpoa_align_sequence_to_graph(abpoa_seam_t *ab, abpoa_seam_para_t *abpt, uint8_t *query, int qlen, abpoa_seam_res_t *res) {
    return simd_abpoa_align_sequence_to_subgraph(ab, abpt, 0 /* ABPOA_SRC_NODE_ID */, 1 /* ABPOA_SINK_NODE_ID */, query, qlen, res);
}

}  // extern "C"
