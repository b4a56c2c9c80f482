// Device-resident partial-order graph for the read-set batch driver (see poa_device.h): graph fusion, row order and the
// "remaining length" of the adaptive band, one wavefront per read-set, graph state resident in HBM between kernels.
//
// Reference behaviour restated here (all tie-breaks and insertion orders matter for consensus identity):
//   fusion            src/abpoa_graph.c:596-672  abpoa_add_subgraph_alignment (match / aligned-node reuse / new node)
//   edges             src/abpoa_graph.c:418-484  abpoa_add_graph_edge (existing edge: weight += w; n_read of the tail node += 1)
//   aligned nodes     src/abpoa_graph.c:377-401
//   remaining length  src/abpoa_graph.c:233-274  (1 + remaining length of the HEAVIEST out-edge's target, first maximum wins)
// What is deliberately NOT the reference's: the row ORDER.  The reference re-runs a Kahn walk over the whole graph before
// every read (abpoa_graph.c:186-231); here the order is maintained incrementally -- every new node is inserted right
// after its predecessor on the alignment path, which keeps the order topological (an alignment path is monotone in row
// order, so no new edge points backwards).  In global mode nothing the DP, the backtrack or the fusion computes depends on
// which topological order is used (predecessor lists keep their in_id order, band and remaining length are functions of
// the graph), and the batch driver's tests check the consensus of every set against the host driver, which keeps the
// reference's order.
#include "poa_bodies.h"      // helpers, poa_prepare_body, poa_fuse_body

namespace abpoa_hip {

// ---------------------------------------------------------------------------------------------------------------------
// round 0: the first read of every set becomes the backbone chain (reference abpoa_add_graph_sequence, :486-502)
__global__ void __launch_bounds__(64) poa_init_kernel(const PoaDev p) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (lane == 0) { st->order_buf = 0; st->n_cells = 0; st->algo_bytes = 0; st->pad = 0; for (int i = 0; i < 4; ++i) st->t_phase[i] = 0; st->algo_bytes_before = 0; }
    if (S.n_reads <= 0) { if (lane == 0) { st->n_nodes = 2; st->status = POA_ST_OK; } return; }
    const int L = p.read_len[S.read0];
    const uint8_t *seq = p.reads + p.read_off[S.read0];
    const int n = L + 2;
    if (n > S.node_cap) { if (lane == 0) { st->n_nodes = 2; st->status = POA_ST_FALLBACK; st->pad = 1; } return; }
    const int64_t N0 = S.node0;
    for (int u = lane; u < n; u += 64) {
        // node ids: 0 = source, 1 = sink, 2 + i = base i of the read
        const int i = u - 2;
        p.nd_base[N0 + u] = u >= 2 ? seq[i] : 0;
        p.nd_naln[N0 + u] = 0;
        int nin = 0, nout = 0, in0 = 0, out0 = 0;
        if (u == 0) { nout = 1; out0 = 2; }
        else if (u == 1) { nin = 1; in0 = L + 1; }
        else { nin = 1; in0 = i == 0 ? 0 : u - 1; nout = 1; out0 = i == L - 1 ? 1 : u + 1; }
        p.nd_nin[N0 + u] = (uint8_t)nin; p.nd_nout[N0 + u] = (uint8_t)nout;
        in_slot(p, N0 + u, 0) = in0; out_slot(p, N0 + u, 0) = out0; outw_slot(p, N0 + u, 0) = 1;
        p.nd_nread[N0 + u] = nout;          // every edge added from a node counts one read through it
        // row order: source, the chain, sink
        const int row = u == 0 ? 0 : (u == 1 ? n - 1 : u - 1);
        p.nd_row[N0 + u] = row; p.row_node[0][N0 + row] = u;
    }
    if (lane == 0) { st->n_nodes = n; st->status = POA_ST_OK; }
}

// ---------------------------------------------------------------------------------------------------------------------
// lock-step rounds: one launch per phase and round, every read-set at round p.round (bodies: poa_bodies.h)
__global__ void __launch_bounds__(GT) poa_prepare_kernel(const PoaDev p) {
    if ((int)blockIdx.x >= p.n_sets) return;
    poa_prepare_body(p, blockIdx.x, p.round);
}
__global__ void __launch_bounds__(GT) poa_fuse_kernel(const PoaDev p) {
    if ((int)blockIdx.x >= p.n_sets) return;
    poa_fuse_body(p, blockIdx.x, p.round);
}

// ---------------------------------------------------------------------------------------------------------------------
// after the last round: heaviest-bundling consensus, reference src/abpoa_output.c:361-415 (scores) and :343-356 (path).
// score[u] = w* + score[t*] where w* is the largest out-edge weight and t* the target with the best score among the
// edges of weight w* (ties: the LAST such edge; for the source: best weight, then best score, ties: the FIRST).  The
// reference fills these in a reverse Kahn walk; they are functions of the successors only, so any reverse topological
// order gives the same values.  Here: 64-row blocks from the sink upwards, lane = row; inside a block a lane fires as soon
// as the in-block targets among its heaviest edges are done (their scores travel through LDS), everything above the block
// is final in HBM.  Then the path is walked from the source, block by block through LDS.
__global__ void __launch_bounds__(64) poa_consensus_kernel(const PoaDev p) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (st->status != POA_ST_OK) return;
    const int n = st->n_nodes;
    if (n <= 2) { if (lane == 0) st->cons_len = 0; return; }
    const int64_t N0 = S.node0;
    const int32_t *order = p.row_node[st->order_buf] + N0;
    int32_t *score = p.row_remain + N0;          // per ROW (pools of the DP inputs, free after the last round)
    int32_t *nextrow = p.row_node_id + N0;       // per row: row of the chosen successor (-1: none)
    __shared__ int sh_score[64], sh_next[64], sh_node[64];
    for (int t0 = ((n - 1) >> 6) << 6; t0 >= 0; t0 -= 64) {
        const int r = t0 + lane; const bool valid = r < n;
        const int u = valid ? order[r] : 1;
        const int no = valid ? (int)p.nd_nout[N0 + u] : 0;
        // heaviest weight and the rows of the candidate targets (edges with that weight), as a dependency mask inside the block
        int wmax = INT_MIN;
        for (int t = 0; t < no; ++t) wmax = imax_(wmax, outw_slot(p, N0 + u, t));
        // candidates = targets of the heaviest edges, in edge order: their rows (and, for targets in later blocks, their final scores)
        // are fetched once, so that a row's turn in the loop below costs LDS reads only; a row with more than NCAND candidates re-reads
        constexpr int NCAND = 4;
        unsigned long long dep = 0; int ctr[NCAND], csc[NCAND], ncand = 0;
#pragma unroll
        for (int c_ = 0; c_ < NCAND; ++c_) { ctr[c_] = -1; csc[c_] = 0; }
        for (int t = 0; t < no; ++t) if (outw_slot(p, N0 + u, t) == wmax) {
            const int tr = p.nd_row[N0 + out_slot(p, N0 + u, t)];
            if (tr < t0 + 64) dep |= 1ull << (tr - t0);
#pragma unroll
            for (int c_ = 0; c_ < NCAND; ++c_) if (c_ == ncand) { ctr[c_] = tr; csc[c_] = tr >= t0 + 64 ? ld_fresh(score + tr) : 0; }
            ++ncand;
        }
        bool done = !valid; int my_score = 0, my_next = -1;
        if (valid && no == 0) { done = true; }                                  // the sink: score 0, no successor
        unsigned long long done_mask = __ballot(done);
        sh_score[lane] = 0; sh_next[lane] = -1;
        __syncthreads();
        for (int it = 0; it < 64 && done_mask != ~0ull; ++it) {
            const bool fire = !done && (dep & ~done_mask) == 0;
            if (fire) {
                int best_sc = INT_MIN, best_row = -1; const bool is_src = u == 0;
                if (ncand <= NCAND) {
#pragma unroll
                    for (int c_ = 0; c_ < NCAND; ++c_) if (c_ < ncand) {
                        const int tr = ctr[c_];
                        const int sc_ = tr < t0 + 64 ? sh_score[tr - t0] : csc[c_];
                        if (best_row < 0 || (is_src ? sc_ > best_sc : sc_ >= best_sc)) { best_sc = sc_; best_row = tr; }
                    }
                } else for (int t = 0; t < no; ++t) if (outw_slot(p, N0 + u, t) == wmax) {
                    const int tr = p.nd_row[N0 + out_slot(p, N0 + u, t)];
                    const int sc_ = tr < t0 + 64 ? sh_score[tr - t0] : ld_fresh(score + tr);
                    if (best_row < 0 || (is_src ? sc_ > best_sc : sc_ >= best_sc)) { best_sc = sc_; best_row = tr; }
                }
                my_score = wmax + best_sc; my_next = best_row; done = true;
                sh_score[lane] = my_score;
            }
            __syncthreads();
            done_mask |= __ballot(fire);
        }
        if (valid) { score[r] = my_score; nextrow[r] = my_next; }
        __syncthreads();
    }
    // ---- path: rows from the source's choice to the sink
    int cur = ld_fresh(nextrow + 0), len = 0; bool overflow = false;
    const int sink_row = n - 1;                   // the sink is always the last row
    while (cur >= 0 && cur < sink_row) {
        const int t0 = cur & ~63;
        const int r = t0 + lane;
        sh_next[lane] = r < n ? ld_fresh(nextrow + r) : -1;
        sh_node[lane] = r < n ? order[r] : 0;
        __syncthreads();
        while (cur >= t0 && cur < t0 + 64 && cur < sink_row) {
            const int u = sh_node[cur - t0];
            if (len < S.cons_cap) { if (lane == 0) { p.cons_node[S.cons0 + len] = u; p.cons_base[S.cons0 + len] = p.nd_base[N0 + u]; p.cons_cov[S.cons0 + len] = p.nd_nread[N0 + u]; } }
            else overflow = true;
            ++len;
            cur = sh_next[cur - t0];
        }
        __syncthreads();
    }
    if (lane == 0) { st->cons_len = len; if (overflow) { st->status = POA_ST_FALLBACK; st->pad = 5; } }
}

static hipError_t launch_k(void (*kern)(const PoaDev), const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(kern, dim3(p.n_sets), dim3(64), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_poa_init(const PoaDev &p, hipStream_t s) { return launch_k(poa_init_kernel, p, s); }
hipError_t launch_poa_prepare(const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(poa_prepare_kernel, dim3(p.n_sets), dim3(GT), (size_t)5 * (size_t)(p.pad > 0 ? p.pad : 0), s, p);      // p.pad: rows the LDS records hold (4 + 1 bytes each; 0: none)
    return hipGetLastError();
}
hipError_t launch_poa_fuse(const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(poa_fuse_kernel, dim3(p.n_sets), dim3(GT), (size_t)16 * GT, s, p);      // (path-exchange records of the fuse body: 4 ints per thread)
    return hipGetLastError();
}
hipError_t launch_poa_consensus(const PoaDev &p, hipStream_t s) { return launch_k(poa_consensus_kernel, p, s); }

}  // namespace abpoa_hip
