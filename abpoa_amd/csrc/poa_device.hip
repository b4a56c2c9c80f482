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
#include <algorithm>
#include <map>
#include <mutex>
#include <utility>
#include "poa_bodies.h"      // helpers, poa_prepare_body, poa_fuse_body

namespace abpoa_hip {

// ---------------------------------------------------------------------------------------------------------------------
// round 0: the first read of every set becomes the backbone chain (reference abpoa_add_graph_sequence, :486-502)
__global__ void __launch_bounds__(64) poa_init_kernel(const PoaDev p) {
    const int s = blockIdx.x, lane = threadIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (lane == 0) { st->order_buf = 0; st->n_cells = 0; st->algo_bytes = 0; st->pad = 0; st->cons_len = 0; st->msa_len = 0; st->cigar_dig = 0; for (int i = 0; i < 4; ++i) st->t_phase[i] = 0;
            st->algo_bytes_before = 0; }
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
        // (edge weight = weight of the base the edge leads to; the edge into the sink takes the last base's: reference abpoa_add_graph_sequence :486-499)
        const int32_t *wq = p.wts ? p.wts + p.read_off[S.read0] : nullptr;
        in_slot(p, S, N0 + u, 0) = in0; out_slot(p, S, N0 + u, 0) = out0; outw_slot(p, S, N0 + u, 0) = !wq ? 1 : (u == 0 ? wq[0] : (u == 1 ? 0 : wq[i == L - 1 ? L - 1 : i + 1]));
        p.nd_nread[N0 + u] = nout;          // every edge added from a node counts one read through it
        // read 0 went through the node's one edge
        if (p.rid_words && nout) { for (int w_ = 0; w_ < p.rid_words; ++w_) p.nd_rid[((N0 + u) * p.out_cap) * p.rid_words + w_] = w_ == 0 ? 1ull : 0ull; }
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
// -s (ambiguous strand), reference abpoa_poa src/abpoa_align.c:315-336.  After the forward alignment of round k: a read whose score is below
// min(qlen, nodes - 2) * max_mat * .3333 is aligned again as its reverse complement (codes 0..3 complemented, every other code 4) on the SAME rows -- the
// reference calls the DP without sorting again, so the adaptive band starts from the bounds the forward run left (DevBatch.fresh_band 0: the general kernel
// on the left / right arrays the forward run wrote).  The check kernel saves the forward result, writes the reverse complement and re-targets the
// descriptor; every other set's descriptor gets ALN_SKIP.  The pick kernel keeps the strand with the strictly better score.
__global__ void __launch_bounds__(GT) poa_strand_check_kernel(const PoaDev p) {
    const int s = blockIdx.x, tid = threadIdx.x, k = p.round;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    if (tid == 0) p.retry[s] = 0;
    AlnDesc d = p.aln[s];
    const AlnOut o = p.out[s];
    const int status = uni(p.state[s].status), flags = uni(d.flags), o_status = uni(o.status), n_cigar = uni(o.n_cigar), qlen = uni(d.qlen), n_rows = uni(d.n_rows), score = uni(o.best_score);
    if (status != POA_ST_OK || k >= S.n_reads || (flags & ALN_SKIP)) return;
    const int lim = qlen < n_rows - 2 ? qlen : n_rows - 2;
    const bool again = o_status == 0 && (double)score < (double)(lim * p.max_mat) * .3333;
    __syncthreads();      // (every thread has its copy of the descriptor and of the result)
    if (!again) { if (tid == 0) p.aln[s].flags = flags | ALN_SKIP; return; }
    const uint64_t *cg = p.cigar + S.cigar_off; uint64_t *cf = p.cigar_fwd + S.cigar_off;
    for (int i = tid; i < n_cigar; i += GT) cf[i] = cg[i];
    const int64_t off = p.read_off[S.read0 + k];
    const uint8_t *q = p.reads + off; uint8_t *rq = p.reads_rc + off;
    for (int j = tid; j < qlen; j += GT) { const uint8_t c = q[qlen - 1 - j]; rq[j] = c < 4 ? (uint8_t)(3 - c) : (uint8_t)4; }
    if (p.wts) for (int j = tid; j < qlen; j += GT) p.wts_rc[off + j] = p.wts[off + qlen - 1 - j];
    if (tid == 0) {
        p.out_fwd[s] = o;
        d.flags = 0; d.query_off = off + (int64_t)(p.reads_rc - p.reads); p.aln[s] = d;
        p.out[s].status = 0; p.out[s].n_cigar = 0; p.out[s].n_cells = 0;
        p.retry[s] = 1;
    }
}
__global__ void __launch_bounds__(GT) poa_strand_pick_kernel(const PoaDev p) {
    const int s = blockIdx.x, tid = threadIdx.x, k = p.round;
    if (s >= p.n_sets || !uni((int)p.retry[s])) return;
    const PoaSet S = p.sets[s];
    const AlnOut rc = p.out[s], fw = p.out_fwd[s];
    const int rc_status = uni(rc.status), rc_score = uni(rc.best_score), fw_score = uni(fw.best_score), fw_cigar = uni(fw.n_cigar);
    __syncthreads();
    if (rc_status != 0) return;      // (the fuse phase sends the set to the host driver)
    if (rc_score > fw_score) { if (tid == 0) { p.is_rc[S.read0 + k] = 1; p.out[s].n_cells = rc.n_cells + fw.n_cells; } return; }
    const uint64_t *cf = p.cigar_fwd + S.cigar_off; uint64_t *cg = p.cigar + S.cigar_off;
    for (int i = tid; i < fw_cigar; i += GT) cg[i] = cf[i];
    if (tid == 0) { AlnOut r = fw; r.n_cells += rc.n_cells; p.out[s] = r; }
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
        for (int t = 0; t < no; ++t) wmax = imax_(wmax, outw_slot(p, S, N0 + u, t));
        // candidates = targets of the heaviest edges, in edge order: their rows (and, for targets in later blocks, their final scores)
        // are fetched once, so that a row's turn in the loop below costs LDS reads only; a row with more than NCAND candidates re-reads
        constexpr int NCAND = 4;
        unsigned long long dep = 0; int ctr[NCAND], csc[NCAND], ncand = 0;
#pragma unroll
        for (int c_ = 0; c_ < NCAND; ++c_) { ctr[c_] = -1; csc[c_] = 0; }
        for (int t = 0; t < no; ++t) if (outw_slot(p, S, N0 + u, t) == wmax) {
            const int tr = p.nd_row[N0 + out_slot(p, S, N0 + u, t)];
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
                } else for (int t = 0; t < no; ++t) if (outw_slot(p, S, N0 + u, t) == wmax) {
                    const int tr = p.nd_row[N0 + out_slot(p, S, N0 + u, t)];
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


// ---------------------------------------------------------------------------------------------------------------------
// The reference's row order, rebuilt before every alignment (order_mode 1): abpoa_BFS_set_node_index, src/abpoa_graph.c:186-231 -- a Kahn walk with a
// FIFO queue in which a node is enqueued only when it AND every node aligned to it have no unvisited predecessor, followed at once by those aligned
// nodes.  Local mode needs exactly this order: the best cell of a local alignment is the first row that reaches the maximum (reference :1012-1016).
//
// One wavefront per read-set.  The queue IS the order array; up to 64 queue entries are processed per pass, a lane each, and the pass reproduces what the
// sequential walk would have done with them.  Every in-degree decrement has a time stamp  key = ORD_KEY * (queue position of the node popped) + (index of
// the out-edge)  -- the order in which the reference performs them.  zt[v] = the largest key that touched v, so once v's count is zero zt[v] is the moment it
// became zero.  The reference enqueues a group at the moment its LAST member reaches zero, that member first and then its aligned list in list order
// (a member that reaches zero earlier fails the check at :212-215 and is not looked at again): so the edge (lane, k) whose key equals zt[v], with every
// aligned node of v at zero and none of them later than v, pushes the group, and the groups of a pass are pushed in key order (prefix sum over the lanes,
// edges in order inside a lane).  A chain graph degenerates to one node per pass: a pass costs one round trip to the node's edge record plus LDS work.
constexpr int ORD_KEY = 256;                  // time stamps: ORD_KEY * queue position + edge index (the source may have up to POA_TERM_MAX out-edges)
constexpr int ORD_RING = 1024;                 // the most recent queue entries, in LDS (the frontier of a POA graph is a few nodes wide)
extern __shared__ int ord_lds[];               // [ORD_RING] ring, then -- LDS tables -- [cap] in-degree counters, [cap] zero times (rank pass: + [cap] rank, [cap] stack)
// table `which`, entry i; n: the stride between tables (LDS: the launch's table capacity; global: the node count)
// (one function per operation, the address space chosen at compile time: a reference picked by `L ? lds : global` is a GENERIC pointer, and every access
//  through it a FLAT instruction -- which counts on both memory counters, so that waiting for a table entry in LDS also waited for the row-order stores of
//  the pass before to be acknowledged by memory, ~1 us per pass)
template <bool L> __device__ __forceinline__ int tbl_ld(const PoaDev &p, int32_t *g, int which, int n, int i) { if constexpr (L) return ord_lds[ORD_RING + which * n + i];
        else return ld_fresh(g + (int64_t)which * n + i); }
template <bool L> __device__ __forceinline__ void tbl_st(const PoaDev &p, int32_t *g, int which, int n, int i, int v) { if constexpr (L) ord_lds[ORD_RING + which * n + i] = v;
        else g[(int64_t)which * n + i] = v; }
template <bool L> __device__ __forceinline__ void tbl_dec_max(const PoaDev &p, int32_t *g, int n, int i, int key) {      // counter (table 0) - 1, zero time (table 1) = max(., key)
    if constexpr (L) { atomicSub(&ord_lds[ORD_RING + i], 1); atomicMax(&ord_lds[ORD_RING + n + i], key); }
    else { atomicSub(g + i, 1); atomicMax(g + (int64_t)n + i, key); }
}

// (Measured and dropped: the adjacency staged in LDS too -- 26 bytes a node.  A walk's wavefront then waits 24 % less, but 70 KB of LDS leave one
//  workgroup per CU instead of six and the launch runs in four turns: 4.1 ms instead of 1.5 ms per launch on configs[4].)
// orders this wavefront's table accesses (one wavefront per read-set: no barrier): LDS tables need the LDS queue drained, global ones the stores acknowledged.
// (Not __syncthreads(): its vmcnt(0) would also wait, every pass, for the stores of the row order -- a memory round trip the walk does not depend on.)
template <bool L> __device__ __forceinline__ void tbl_fence() {
    if (L) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
template <bool L>
__device__ __forceinline__ bool poa_order_body(const PoaDev &p, const PoaSet &S, const int n, int32_t *order) {
    const int lane = threadIdx.x;
    const int64_t N0 = S.node0;
    int32_t *g = p.scratch + S.scratch0;           // (global tables when the graph is larger than the LDS tables: [n] counters, [n] zero times)
    // (LDS tables: the size of every node's aligned group as a byte table behind them -- the group size of a TARGET was the second dependent memory round
    //  trip of a pass, after the popped node's out-edge record)
    // ... and, 16 bits a node, the target of a node's ONLY out-edge (0xffff: none or several, or a graph of 65535+ nodes: the edge record comes from memory):
    // most nodes of a POA graph are links of a chain, and the out-edge record of the node just popped was the one memory round trip (~1 us) left in a pass
    uint8_t *const s_naln = (uint8_t *)(ord_lds + ORD_RING + 2 * n);
    uint16_t *const s_next = (uint16_t *)(ord_lds + ORD_RING + 2 * n + ((n + 3) >> 2));
    // ... and the QUEUE itself, 16 bits an entry, for graphs of fewer than 65535 nodes: the row order and nd_row go to memory in one coalesced sweep after
    // the walk.  (Stored pass by pass they kept a store in flight at every wait for a load: vmcnt counts loads and stores alike, and a store takes ~1 us to be
    // acknowledged -- measured: 3000 cycles per single-node pass, 5600 per parallel pass, nearly all of it that.)
    uint16_t *const s_q = s_next + ((n + 1) & ~1);
    const bool QL = L && n < 65535;
    auto qget = [&](int pos_, int head_, int tail_) __attribute__((always_inline)) -> int {
        if (QL) return (int)s_q[pos_];
        if (tail_ - head_ <= ORD_RING) return ord_lds[pos_ & (ORD_RING - 1)];
        return ld_fresh(order + pos_);
    };
    auto qput = [&](int at_, int node_) __attribute__((always_inline)) {
        if (QL) s_q[at_] = (uint16_t)node_;
        else { order[at_] = node_; p.nd_row[N0 + node_] = at_; ord_lds[at_ & (ORD_RING - 1)] = node_; }
    };
    for (int u = lane; u < n; u += 64) {
        tbl_st<L>(p, g, 0, n, u, p.nd_nin[N0 + u]); tbl_st<L>(p, g, 1, n, u, -1);
        if (L) { s_naln[u] = p.nd_naln[N0 + u]; s_next[u] = (p.nd_nout[N0 + u] == 1 && n < 65535) ? (uint16_t)p.nd_out[(N0 + u) * POA_HOT] : (uint16_t)0xffff; }
    }
    auto naln_of = [&](int v) __attribute__((always_inline)) -> int { return L ? (int)s_naln[v] : (int)p.nd_naln[N0 + v]; };
    if (lane == 0) qput(0, 0);
    tbl_fence<L>();
    int head = 0, tail = 1;
#ifdef ABPOA_HIP_ORDER_PROF
    long long t_seq = 0, t_par = 0, n_seq = 0, n_par = 0, t_last = (long long)__builtin_amdgcn_s_memtime();
#define ORD_PROF(T, N) { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); T += t_ - t_last; t_last = t_; ++N; }
#else
#define ORD_PROF(T, N)
#endif
    while (head < tail) {
        if (tail - head == 1) {
            // ONE node in the queue (most passes: a POA graph is chains with short bubbles): the reference's own sequential step, executed by every lane
            // with the same values (wave-uniform control flow, broadcast LDS reads; lane 0 stores) -- no atomics, no time-stamp comparison, no scan:
            // a third of the LDS round trips of the parallel pass below.  zt is kept up to date for the passes that compare it.
            if (!QL && tail - head > ORD_RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int u = uni(qget(head, head, tail));
            int no; int4 o4;
            const int nx = L ? uni((int)s_next[u]) : 0xffff;
            if (nx != 0xffff) { no = 1; o4 = make_int4(nx, 0, 0, 0); }
            else { no = uni((int)p.nd_nout[N0 + u]); o4 = *(const int4 *)(p.nd_out + (N0 + u) * POA_HOT); }
            int at = tail;
            for (int k = 0; k < no; ++k) {
                const int v = uni(k == 0 ? o4.x : (k == 1 ? o4.y : (k == 2 ? o4.z : (k == 3 ? o4.w : out_slot(p, S, N0 + u, k)))));
                const int d = uni(tbl_ld<L>(p, g, 0, n, v)) - 1;
                if (lane == 0) { tbl_st<L>(p, g, 0, n, v, d); tbl_st<L>(p, g, 1, n, v, head * ORD_KEY + k); }
                if (d != 0) continue;
                const int na = uni(naln_of(v));
                bool ready = true;
                // (fence: lane 0's store above may be what an aligned node's entry holds)
                if (na > 0) { tbl_fence<L>(); for (int t = 0; t < na && ready; ++t) ready = uni(tbl_ld<L>(p, g, 0, n, p.nd_aln[(N0 + v) * p.aln_cap + t])) == 0; }
                if (!ready) continue;
                if (at + 1 + na > n) return false;
                if (lane == 0) qput(at, v);
                if (lane < na) qput(at + 1 + lane, p.nd_aln[(N0 + v) * p.aln_cap + lane]);
                at += 1 + na;
            }
            tail = at; ++head;
            tbl_fence<L>();
            ORD_PROF(t_seq, n_seq)
            continue;
        }
        const int cnt = imin_(64, tail - head), pos = head + lane; const bool act = lane < cnt;
        int u = 1, no = 0; int4 o4 = make_int4(0, 0, 0, 0);
        if (!QL && tail - head > ORD_RING) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (a frontier wider than the ring: from the order array, once its stores are acknowledged)
        if (act) u = qget(pos, head, tail);
        if (act) { no = p.nd_nout[N0 + u]; o4 = *(const int4 *)(p.nd_out + (N0 + u) * POA_HOT); }
        auto target = [&](int k) { return k == 0 ? o4.x : (k == 1 ? o4.y : (k == 2 ? o4.z : (k == 3 ? o4.w : out_slot(p, S, N0 + u, k)))); };
        for (int k = 0; k < no; ++k) tbl_dec_max<L>(p, g, n, target(k), pos * ORD_KEY + k);
        tbl_fence<L>();
        // Which edges push a group.  The first four edges of a lane and the first four aligned nodes of their targets are looked at together, so that the
        // memory loads of this half of the pass -- the aligned lists -- are all in flight at once (as a loop with an early exit they were one dependent
        // round trip per aligned node: 78 % of the columns of a 30-sequence protein MSA hold a group); longer lists take the loop.
        constexpr int KQ = 4, TQ = 4;
        unsigned trig = 0; int total = 0;
        int vq[KQ], naq[KQ], aq[KQ][TQ]; bool cq[KQ];
#pragma unroll
        for (int k = 0; k < KQ; ++k) {
            vq[k] = k < no ? target(k) : 0;
            cq[k] = k < no && tbl_ld<L>(p, g, 0, n, vq[k]) == 0 && tbl_ld<L>(p, g, 1, n, vq[k]) == pos * ORD_KEY + k;
            naq[k] = cq[k] ? naln_of(vq[k]) : 0;
        }
#pragma unroll
        for (int k = 0; k < KQ; ++k)
#pragma unroll
            for (int t = 0; t < TQ; ++t) aq[k][t] = (cq[k] && t < naq[k]) ? p.nd_aln[(N0 + vq[k]) * p.aln_cap + t] : 0;
#pragma unroll
        for (int k = 0; k < KQ; ++k) if (cq[k]) {
            const int key = pos * ORD_KEY + k; bool ready = true;
#pragma unroll
            for (int t = 0; t < TQ; ++t) if (t < naq[k]) ready = ready && tbl_ld<L>(p, g, 0, n, aq[k][t]) == 0 && tbl_ld<L>(p, g, 1, n, aq[k][t]) < key;
            for (int t = TQ; t < naq[k] && ready; ++t) { const int a = p.nd_aln[(N0 + vq[k]) * p.aln_cap + t]; ready = tbl_ld<L>(p, g, 0, n, a) == 0 && tbl_ld<L>(p, g, 1, n, a) < key; }
            if (ready) { trig |= 1u << k; total += 1 + naq[k]; }
        }
        for (int k = KQ; k < no; ++k) {
            const int v = target(k), key = pos * ORD_KEY + k;
            if (tbl_ld<L>(p, g, 0, n, v) != 0 || tbl_ld<L>(p, g, 1, n, v) != key) continue;
            const int na = naln_of(v); bool ready = true;
            for (int t = 0; t < na && ready; ++t) { const int a = p.nd_aln[(N0 + v) * p.aln_cap + t]; ready = tbl_ld<L>(p, g, 0, n, a) == 0 && tbl_ld<L>(p, g, 1, n, a) < key; }
            if (ready) { trig |= 1u << k; total += 1 + na; }
        }
        const int incl = wave_scan_add(total), all = __builtin_amdgcn_readlane(incl, 63);
        if (tail + all > n) return false;                                          // (more entries than nodes: not a graph this walk understands)
        int at = tail + incl - total;
        auto put = [&](int node_) __attribute__((always_inline)) { qput(at, node_); ++at; };
#pragma unroll
        for (int k = 0; k < KQ; ++k) if (trig >> k & 1) {
            put(vq[k]);
#pragma unroll
            for (int t = 0; t < TQ; ++t) if (t < naq[k]) put(aq[k][t]);
            for (int t = TQ; t < naq[k]; ++t) put(p.nd_aln[(N0 + vq[k]) * p.aln_cap + t]);
        }
        for (int k = KQ; k < no; ++k) if (trig >> k & 1) {
            const int v = target(k), na = naln_of(v);
            put(v);
            for (int t = 0; t < na; ++t) put(p.nd_aln[(N0 + v) * p.aln_cap + t]);
        }
        tail += all; head += cnt;
        tbl_fence<L>();
        ORD_PROF(t_par, n_par)
    }
    if (QL) {      // the order and every node's row, in one sweep
        if (head != n || s_q[n - 1] != 1) return false;
        for (int r = lane; r < n; r += 64) { const int u_ = s_q[r]; order[r] = u_; p.nd_row[N0 + u_] = r; }
    }
#ifdef ABPOA_HIP_ORDER_PROF
    if (lane == 0) { PoaState *st_ = p.state + (int)blockIdx.x; st_->t_phase[0] += t_seq; st_->t_phase[1] += t_par; st_->t_phase[2] += n_seq; st_->t_phase[3] += n_par; }
#endif
    return head == n;
}


// ---- the same walk with EVERYTHING it touches in LDS (graphs of up to PoaDev.order_lds nodes whose aligned lists hold at most PoaDev.order_ecap entries:
//      what BASELINE.json configs[4] and every short-read MSA look like).  Measured on the version above: a pass costs 3000 (one node) to 5600 (several)
//      cycles, nearly all of it memory round trips -- the popped node's out-edge record, the aligned lists of every candidate, and, through vmcnt, the
//      acknowledgement of the order stores of the pass before.  Here:
//   * ONE counter per aligned GROUP instead of one per node: cnt[rep] = sum of the members' in-degrees (rep = the member with the smallest id).  The
//     reference enqueues a group when the decrement that zeroes a member finds every other member at zero (abpoa_graph.c:210-221) -- that is the LAST
//     decrement over all in-edges of the group, i.e. the one that brings the sum to zero; its target is the member pushed first.  No list is read to decide;
//   * among the edges of one pass that hit a group whose sum is now zero the latest in the reference's order (key = 16 x queue position + edge index) wins:
//     the counter word, free once it is zero, takes the maximum of key + 1;
//   * aligned lists as a CSR of 16-bit ids (list order kept: it is the order in which the members follow the trigger into the queue), the only out-edge of
//     a chain node, the group sizes and the queue itself: all LDS; the out-edge record of a node with several edges is the one memory read left.
//   The row order and nd_row are written in one coalesced sweep at the end.
__device__ __forceinline__ int poa_order_body_lds(const PoaDev &p, const PoaSet &S, const int n, int32_t *order) {      // 1 done, 0 failed, -1: does not fit (the caller takes the general body)
    const int lane = threadIdx.x;
    const int64_t N0 = S.node0;
    int *const cnt = ord_lds;                                            // [n]
    uint16_t *const s_rep = (uint16_t *)(cnt + n), *const s_next = s_rep + n, *const s_q = s_next + n, *const s_aoff = s_q + n;      // [n] each, s_aoff [n + 1]
    uint16_t *const s_alist = s_aoff + n + 1 + (n & 1 ? 0 : 1);         // [order_ecap]
    uint8_t *const s_naln = (uint8_t *)(s_alist + p.order_ecap);         // [n]
    // ---- tables: group sizes and CSR offsets (prefix sum in node order), then representatives, counters, lists, chain links
    int carry = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int u = t0 + lane, na = u < n ? (int)p.nd_naln[N0 + u] : 0;
        const int incl = wave_scan_add(na);
        if (u < n) { s_naln[u] = (uint8_t)na; s_aoff[u] = (uint16_t)imin_(carry + incl - na, 65535); cnt[u] = 0; }
        carry += __builtin_amdgcn_readlane(incl, 63);
    }
    if (carry > p.order_ecap || carry > 65535) return -1;
    if (lane == 0) s_aoff[n] = (uint16_t)carry;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    for (int u = lane; u < n; u += 64) {
        const int na = s_naln[u], off = s_aoff[u]; int r = u;
        for (int t = 0; t < na; ++t) { const int a = p.nd_aln[(N0 + u) * p.aln_cap + t]; s_alist[off + t] = (uint16_t)a; r = imin_(r, a); }
        s_rep[u] = (uint16_t)r;
        atomicAdd(&cnt[r], (int)p.nd_nin[N0 + u]);
        s_next[u] = p.nd_nout[N0 + u] == 1 ? (uint16_t)p.nd_out[(N0 + u) * POA_HOT] : (uint16_t)0xffff;
    }
    if (lane == 0) s_q[0] = 0;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier();
    auto fence = []() __attribute__((always_inline)) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_wave_barrier(); };
    int head = 0, tail = 1;
    while (head < tail) {
        if (tail - head == 1) {      // one node in the queue: the reference's own step on wave-uniform values (lane 0 stores; a pushed group's members a lane each)
            const int u = uni((int)s_q[head]);
            int no; int4 o4;
            const int nx = uni((int)s_next[u]);
            if (nx != 0xffff) { no = 1; o4 = make_int4(nx, 0, 0, 0); }
            else { no = uni((int)p.nd_nout[N0 + u]); o4 = *(const int4 *)(p.nd_out + (N0 + u) * POA_HOT); }
            int at = tail;
            for (int k = 0; k < no; ++k) {
                const int v = uni(k == 0 ? o4.x : (k == 1 ? o4.y : (k == 2 ? o4.z : (k == 3 ? o4.w : out_slot(p, S, N0 + u, k)))));
                const int r = uni((int)s_rep[v]);
                const int d = uni(cnt[r]) - 1;
                if (lane == 0) cnt[r] = d != 0 ? d : 0x7fffffff;          // (a group that was pushed keeps a value no decrement reaches)
                if (d != 0) { fence(); continue; }                        // (the next edge may hit the same group)
                const int na = uni((int)s_naln[v]), off = uni((int)s_aoff[v]);
                if (at + 1 + na > n) return 0;
                if (lane == 0) s_q[at] = (uint16_t)v;
                if (lane < na) s_q[at + 1 + lane] = s_alist[off + lane];
                at += 1 + na;
                fence();
            }
            tail = at; ++head;
            fence();
            continue;
        }
        const int cnt_ = imin_(64, tail - head), pos = head + lane; const bool act = lane < cnt_;
        int u = 1, no = 0; int4 o4 = make_int4(0, 0, 0, 0);
        if (act) {
            u = s_q[pos];
            const int nx = s_next[u];
            if (nx != 0xffff) { no = 1; o4.x = nx; } else { no = p.nd_nout[N0 + u]; o4 = *(const int4 *)(p.nd_out + (N0 + u) * POA_HOT); }
        }
        auto target = [&](int k) { return k == 0 ? o4.x : (k == 1 ? o4.y : (k == 2 ? o4.z : (k == 3 ? o4.w : out_slot(p, S, N0 + u, k)))); };
        for (int k = 0; k < no; ++k) atomicSub(&cnt[s_rep[target(k)]], 1);
        fence();
        unsigned zero = 0;                                               // edges whose group's sum is zero now
        for (int k = 0; k < no; ++k) if (cnt[s_rep[target(k)]] == 0) zero |= 1u << k;
        fence();
        for (int k = 0; k < no; ++k) if (zero >> k & 1) atomicMax(&cnt[s_rep[target(k)]], pos * 16 + k + 1);
        fence();
        unsigned trig = 0; int total = 0;
        for (int k = 0; k < no; ++k) if ((zero >> k & 1) && cnt[s_rep[target(k)]] == pos * 16 + k + 1) { trig |= 1u << k; total += 1 + s_naln[target(k)]; }
        const int incl = wave_scan_add(total), all = __builtin_amdgcn_readlane(incl, 63);
        if (tail + all > n) return 0;
        int at = tail + incl - total;
        for (int k = 0; k < no; ++k) if (trig >> k & 1) {
            const int v = target(k), na = s_naln[v], off = s_aoff[v];
            s_q[at++] = (uint16_t)v;
            for (int t = 0; t < na; ++t) s_q[at++] = s_alist[off + t];
            cnt[s_rep[v]] = 0x7fffffff;
        }
        tail += all; head += cnt_;
        fence();
    }
    if (head != n || s_q[n - 1] != 1) return 0;
    for (int r = lane; r < n; r += 64) { const int u_ = s_q[r]; order[r] = u_; p.nd_row[N0 + u_] = r; }
    return 1;
}

__global__ void __launch_bounds__(64) poa_order_kernel(const PoaDev p) {
    const int s = blockIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (uni(st->status) != POA_ST_OK || p.round >= S.n_reads) return;
    const int n = uni(st->n_nodes);
    if (n <= 2) return;
    int32_t *order = p.row_node[uni(st->order_buf)] + S.node0;
    int done = -1;
    if (n <= p.order_lds && n < 65535 && p.order_ecap > 0) done = poa_order_body_lds(p, S, n, order);
    const bool ok = done >= 0 ? done == 1 : (n <= p.order_lds ? poa_order_body<true>(p, S, n, order) : poa_order_body<false>(p, S, n, order));
    // (the sink is the last node the walk reaches, reference :203-206; anything else means the graph is not what the fuse phase should have left)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0 && (!ok || ld_fresh(order + n - 1) != 1)) { st->status = POA_ST_FALLBACK; st->pad = 7; }
}

// ---------------------------------------------------------------------------------------------------------------------
// MSA output, pass 1: one column per aligned group, numbered by the reference's depth-first variant of the same walk (abpoa_DFS_set_msa_rank,
// src/abpoa_graph.c:315-362: a LIFO stack instead of the queue; a node takes the next rank when it is popped, together with its aligned nodes, unless
// one of them was popped before).  A stack walk is sequential in the nodes; the out-edges of the popped node are handled a lane each, with the same
// time-stamp rule as above for which edge pushes a group (two out-edges of one node may lead to two members of one group: the later edge pushes).
template <bool L>
__device__ __forceinline__ int poa_msa_rank_body(const PoaDev &p, const PoaSet &S, const int n) {
    const int lane = threadIdx.x;
    const int64_t N0 = S.node0;
    int32_t *g = p.scratch + S.scratch0;           // tables: 0 counters, 1 zero times, 2 rank, 3 stack
    for (int u = lane; u < n; u += 64) { tbl_st<L>(p, g, 0, n, u, p.nd_nin[N0 + u]); tbl_st<L>(p, g, 1, n, u, -1); tbl_st<L>(p, g, 2, n, u, 0); }
    if (lane == 0) { tbl_st<L>(p, g, 3, n, 0, 0); tbl_st<L>(p, g, 2, n, 0, -1); }
    __syncthreads();
    int sp = 1, msa_rank = 0, pops = 0; bool done = false;
    while (sp > 0 && !done) {
        const int cur = uni(tbl_ld<L>(p, g, 3, n, sp - 1)); --sp;
        const int na_c = p.nd_naln[N0 + cur], no = cur == 1 ? 0 : (int)p.nd_nout[N0 + cur];
        if (uni(tbl_ld<L>(p, g, 2, n, cur)) < 0) {
            if (lane == 0) tbl_st<L>(p, g, 2, n, cur, msa_rank);
            if (lane < na_c) tbl_st<L>(p, g, 2, n, p.nd_aln[(N0 + cur) * p.aln_cap + lane], msa_rank);      // (aln_cap <= 26 < 64: a lane each)
            ++msa_rank;
        }
        if (cur == 1) { done = true; break; }
        // (64 out-edges at a time, a lane each -- only the source can have more: PoaSet.term0 -- chunk after chunk as the sequential walk would take them)
        for (int e0 = 0; e0 < no; e0 += 64) {
            const int e = e0 + lane, key = pops * ORD_KEY + e;
            int v = -1;
            if (e < no) { v = out_slot(p, S, N0 + cur, e); tbl_dec_max<L>(p, g, n, v, key); }
            __syncthreads();
            int total = 0, na = 0;
            if (e < no && tbl_ld<L>(p, g, 0, n, v) == 0) {
                na = p.nd_naln[N0 + v]; bool ready = true;
                for (int t = 0; t < na && ready; ++t) { const int a = p.nd_aln[(N0 + v) * p.aln_cap + t]; ready = tbl_ld<L>(p, g, 0, n, a) == 0 && tbl_ld<L>(p, g, 1, n, a) < key; }
                if (ready) total = 1 + na;
            }
            const int incl = wave_scan_add(total), all = __builtin_amdgcn_readlane(incl, 63);
            if (sp + all > n) return -1;
            if (total) {
                int at = sp + incl - total;
                tbl_st<L>(p, g, 3, n, at, v); tbl_st<L>(p, g, 2, n, v, -1); ++at;
                for (int t = 0; t < na; ++t) { const int a = p.nd_aln[(N0 + v) * p.aln_cap + t]; tbl_st<L>(p, g, 3, n, at, a); tbl_st<L>(p, g, 2, n, a, -1); ++at; }
            }
            sp += all;
            __syncthreads();
        }
        ++pops;
    }
    if (!done) return -1;
    __syncthreads();
    // the column of a node: the largest rank in its group (abpoa_output.c:141-147; the members of a group share their rank)
    for (int u = lane; u < n; u += 64) {
        int rk = tbl_ld<L>(p, g, 2, n, u); const int na = p.nd_naln[N0 + u];
        for (int t = 0; t < na; ++t) rk = imax_(rk, tbl_ld<L>(p, g, 2, n, p.nd_aln[(N0 + u) * p.aln_cap + t]));
        p.msa_rank[N0 + u] = rk;
    }
    return uni(tbl_ld<L>(p, g, 2, n, 1)) - 1;                                       // msa_len = rank of the sink - 1 (abpoa_output.c:130)
}

__global__ void __launch_bounds__(64) poa_msa_rank_kernel(const PoaDev p) {
    const int s = blockIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    PoaState *st = p.state + s;
    if (uni(st->status) != POA_ST_OK) return;
    const int n = uni(st->n_nodes);
    if (n <= 2) { if (threadIdx.x == 0) st->msa_len = 0; return; }
    const int len = 4 * n <= 2 * p.order_lds ? poa_msa_rank_body<true>(p, S, n) : poa_msa_rank_body<false>(p, S, n);      // (four tables in the space of the order kernel's two)
    if (threadIdx.x == 0) { if (len < 0) { st->status = POA_ST_FALLBACK; st->pad = 8; st->msa_len = 0; } else st->msa_len = len; }
}

// MSA output, pass 2 (after the host has laid the sets' results out back to back: msa_off): rows of gap codes, then every node writes its base into the
// column of its group for each read that left it through one of its out-edges (abpoa_set_msa_seq, abpoa_output.c:103-120); the consensus row likewise
// from the consensus path (:151-164).  Two nodes of one column belong to one aligned group and a read passes through one of them: no two writers per cell.
__global__ void __launch_bounds__(GT) poa_msa_fill_kernel(const PoaDev p) {
    const int s = blockIdx.x, tid = threadIdx.x;
    if (s >= p.n_sets) return;
    const PoaSet S = p.sets[s];
    const PoaState *st = p.state + s;
    if (st->status != POA_ST_OK) return;
    const int n = st->n_nodes, len = st->msa_len, rows = S.n_reads + (p.msa_cons ? 1 : 0);
    if (n <= 2 || len <= 0) return;
    uint8_t *out = p.msa_out + p.msa_off[s];
    const int64_t N0 = S.node0, cells = (int64_t)rows * len;
    for (int64_t i = tid; i < cells; i += GT) out[i] = (uint8_t)p.m;
    __syncthreads();
    for (int u = 2 + tid; u < n; u += GT) {
        const int col = p.msa_rank[N0 + u] - 1, no = p.nd_nout[N0 + u]; const uint8_t b = p.nd_base[N0 + u];
        for (int e = 0; e < no; ++e) for (int w_ = 0; w_ < p.rid_words; ++w_) {
            unsigned long long num = p.nd_rid[((N0 + u) * p.out_cap + e) * p.rid_words + w_];
            while (num) { const int r = w_ * 64 + __builtin_ctzll(num); num &= num - 1; if (r < S.n_reads) out[(int64_t)r * len + col] = b; }
        }
    }
    if (p.msa_cons) for (int i = tid; i < st->cons_len; i += GT) out[(int64_t)S.n_reads * len + p.msa_rank[N0 + p.cons_node[S.cons0 + i]] - 1] = p.cons_base[S.cons0 + i];
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
hipError_t launch_poa_strand_check(const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(poa_strand_check_kernel, dim3(p.n_sets), dim3(GT), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_poa_strand_pick(const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(poa_strand_pick_kernel, dim3(p.n_sets), dim3(GT), 0, s, p);
    return hipGetLastError();
}
hipError_t launch_poa_consensus(const PoaDev &p, hipStream_t s) { return launch_k(poa_consensus_kernel, p, s); }
// the larger of the two layouts: the general walk's ring + two int tables + byte and 16-bit tables; the all-in-LDS walk's 13 bytes a node + 2 per aligned-list entry
size_t poa_order_lds_bytes(int node_cap, int ecap) {
    const size_t nc = (size_t)(node_cap > 0 ? node_cap : 0);
    return std::max(4 * (size_t)ORD_RING + 13 * nc + 32, 13 * nc + 2 * (size_t)(ecap > 0 ? ecap : 0) + 64);
}
static hipError_t launch_ord(void (*kern)(const PoaDev), const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    const size_t lds = poa_order_lds_bytes(p.order_lds, p.order_ecap);
    // (above 64 KB the kernel's dynamic-LDS limit has to be raised: once per kernel, device and size -- the call is slow enough to stall a queue of
    //  back-to-back launches when repeated every round)
    if (lds > 65536) {
        static std::mutex mu; static std::map<std::pair<const void *, int>, size_t> raised;
        int dev = 0; (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> lk(mu);
        size_t &have = raised[{(const void *)kern, dev}];
        if (have < lds) { hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); if (e != hipSuccess) return e; have = lds; }
    }
    hipLaunchKernelGGL(kern, dim3(p.n_sets), dim3(64), lds, s, p);
    return hipGetLastError();
}
hipError_t launch_poa_order(const PoaDev &p, hipStream_t s) { return launch_ord(poa_order_kernel, p, s); }
hipError_t launch_poa_msa_rank(const PoaDev &p, hipStream_t s) { return launch_ord(poa_msa_rank_kernel, p, s); }
hipError_t launch_poa_msa_fill(const PoaDev &p, hipStream_t s) {
    if (p.n_sets <= 0) return hipSuccess;
    hipLaunchKernelGGL(poa_msa_fill_kernel, dim3(p.n_sets), dim3(GT), 0, s, p);
    return hipGetLastError();
}

}  // namespace abpoa_hip
