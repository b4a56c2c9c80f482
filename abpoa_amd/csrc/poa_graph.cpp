// Host partial-order graph: see poa_graph.h for the reference line ranges each routine is pinned to.
#include "poa_graph.h"
#include <limits.h>
#include <string.h>
#include <stdexcept>

namespace abpoa_hip {

abpoa_hip_problem_t FlatProblem::view(const uint8_t *query, int qlen) {
    abpoa_hip_problem_t p;
    p.n_rows = (int)row_base.size(); p.qlen = qlen; p.query = query;
    p.row_base = row_base.data(); p.row_node_id = row_node_id.data();
    p.row_remain = row_remain.empty() ? nullptr : row_remain.data();
    p.row_active = nullptr;
    p.pred_off = pred_off.data(); p.pred_row = pred_row.data(); p.out_off = out_off.data(); p.out_row = out_row.data();
    p.max_pos_left = left.empty() ? nullptr : left.data(); p.max_pos_right = right.empty() ? nullptr : right.data();
    return p;
}

void PoaGraph::reset(int tot_reads, bool use_read_ids) {
    nodes_.clear(); nodes_.resize(2);
    read_ids_.clear();
    tot_reads_ = tot_reads; use_read_ids_ = use_read_ids;
    words_ = use_read_ids ? 1 + ((tot_reads > 0 ? tot_reads : 1) - 1) / 64 : 0;   // reference abpoa_graph.c:599
    if (use_read_ids_) read_ids_.resize(2);
    sorted_ = false; n_edges_ = 0;
}

void PoaGraph::import_nodes(int n, const uint8_t *base, const uint8_t *nin, const uint8_t *nout, const uint8_t *naln, const int32_t *in_id, int in_cap,
                            const int32_t *out_id, const int32_t *out_w, int out_cap, const int32_t *aligned, int aln_cap, const int32_t *n_read) {
    nodes_.clear(); nodes_.resize(n); read_ids_.clear(); use_read_ids_ = false; words_ = 0; n_edges_ = 0; sorted_ = false;
    for (int u = 0; u < n; ++u) {
        PoaNode &nd = nodes_[u];
        nd.base = base[u]; nd.n_read = n_read[u];
        for (int t = 0; t < nin[u]; ++t) nd.in_id.push_back(in_id[(size_t)u * in_cap + t]);
        for (int t = 0; t < nout[u]; ++t) { nd.out_id.push_back(out_id[(size_t)u * out_cap + t]); nd.out_w.push_back(out_w[(size_t)u * out_cap + t]); ++n_edges_; }
        for (int t = 0; t < naln[u]; ++t) nd.aligned.push_back(aligned[(size_t)u * aln_cap + t]);
    }
}

int PoaGraph::add_node(uint8_t base) {                        // reference abpoa_graph.c:409-416
    nodes_.emplace_back(); nodes_.back().base = base;
    if (use_read_ids_) read_ids_.emplace_back();
    return (int)nodes_.size() - 1;
}

void PoaGraph::add_edge(int from, int to, bool check_edge, int w, bool add_read_id, int read_id) {   // :418-484
    PoaNode &f = nodes_[from];
    int edge_i = -1;
    if (check_edge) {
        for (int i = 0; i < f.out_id.size(); ++i) if (f.out_id[i] == to) { f.out_w[i] += w; edge_i = i; break; }
    }
    if (edge_i < 0) {
        nodes_[to].in_id.push_back(from);
        f.out_id.push_back(to); f.out_w.push_back(w); ++n_edges_;
        edge_i = f.out_id.size() - 1;
        if (use_read_ids_) read_ids_[from].resize((size_t)f.out_id.size() * words_, 0);
    }
    if (add_read_id && use_read_ids_) read_ids_[from][(size_t)edge_i * words_ + read_id / 64] |= 1ULL << (read_id & 63);
    f.n_read += 1;
}

int PoaGraph::aligned_with_base(int node_id, uint8_t base) const {      // :377-386
    const PoaNode &n = nodes_[node_id];
    for (int i = 0; i < n.aligned.size(); ++i) if (nodes_[n.aligned[i]].base == base) return n.aligned[i];
    return -1;
}

void PoaGraph::add_aligned(int node_id, int new_id) {                  // :393-401
    const int na = nodes_[node_id].aligned.size();
    for (int i = 0; i < na; ++i) {
        int other = nodes_[node_id].aligned[i];
        nodes_[other].aligned.push_back(new_id);
        nodes_[new_id].aligned.push_back(other);
    }
    nodes_[node_id].aligned.push_back(new_id);
    nodes_[new_id].aligned.push_back(node_id);
}

void PoaGraph::add_alignment(const uint8_t *seq, int len, const uint64_t *cigar, int n_cigar, int read_id, const int32_t *weight) {
    const bool rid = use_read_ids_;
    auto w = [&](int q) { return weight ? (int)weight[q] : 1; };      // reference: weight[query_id] on every edge INTO the base's node, :486-499 / :634-667
    if (nodes_.size() == 2) {                                          // empty graph: :486-502
        if (len <= 0) throw std::invalid_argument("empty first read");
        int last = SRC;
        for (int i = 0; i < len; ++i) { int cur = add_node(seq[i]); add_edge(last, cur, false, w(i), rid, read_id); last = cur; }
        add_edge(last, SINK, false, w(len - 1), rid, read_id);
        sorted_ = false; return;
    }
    if (n_cigar == 0) return;                                          // :614-616
    int query_id = -1, last_id = SRC; bool last_new = false;
    for (int i = 0; i < n_cigar; ++i) {                                // :625-664
        const int op = (int)(cigar[i] & 0xf);
        if (op == ABPOA_HIP_CMATCH) {
            const int node_id = (int)((cigar[i] >> 34) & 0x3fffffff);
            ++query_id;
            if (nodes_[node_id].base != seq[query_id]) {
                int al = aligned_with_base(node_id, seq[query_id]);
                if (al != -1) { add_edge(last_id, al, !last_new, w(query_id), rid, read_id); last_id = al; last_new = false; }
                else {
                    int nid = add_node(seq[query_id]);
                    add_edge(last_id, nid, false, w(query_id), rid, read_id);
                    last_id = nid; last_new = true;
                    add_aligned(node_id, nid);
                }
            } else { add_edge(last_id, node_id, !last_new, w(query_id), rid, read_id); last_id = node_id; last_new = false; }
        } else if (op == ABPOA_HIP_CINS || op == 4 || op == 5) {       // insertion / clips add nodes
            const int l = (int)((cigar[i] >> 4) & 0x3fffffff);
            query_id += l;
            for (int j = l - 1; j >= 0; --j) {
                int nid = add_node(seq[query_id - j]);
                add_edge(last_id, nid, false, w(query_id - j), rid, read_id);
                last_id = nid; last_new = true;
            }
        }                                                              // deletion: nothing
    }
    add_edge(last_id, SINK, !last_new, w(len - 1), rid, read_id);       // :667
    sorted_ = false;
}

void PoaGraph::topological_sort(bool with_remain) {
    const int n = (int)nodes_.size();
    index_to_node_.assign(n, -1); node_to_index_.assign(n, -1);
    std::vector<int> &deg = scratch_deg_, &q = scratch_q_;
    deg.resize(n); q.clear(); q.reserve(n);
    for (int i = 0; i < n; ++i) deg[i] = nodes_[i].in_id.size();
    // Kahn order with aligned groups kept adjacent, reference abpoa_graph.c:186-231
    q.push_back(SRC);
    size_t head = 0; int index = 0; bool done = false;
    while (head < q.size()) {
        const int cur = q[head++];
        index_to_node_[index] = cur; node_to_index_[cur] = index++;
        if (cur == SINK) { done = true; break; }
        const PoaNode &c = nodes_[cur];
        for (int i = 0; i < c.out_id.size(); ++i) {
            const int o = c.out_id[i];
            if (--deg[o] != 0) continue;
            const PoaNode &on = nodes_[o];
            bool ready = true;
            for (int j = 0; j < on.aligned.size(); ++j) if (deg[on.aligned[j]] != 0) { ready = false; break; }
            if (!ready) continue;
            q.push_back(o);
            for (int j = 0; j < on.aligned.size(); ++j) q.push_back(on.aligned[j]);
        }
    }
    if (!done || index != n) throw std::runtime_error("topological sort failed (graph not connected to sink)");
    if (with_remain) {
        // distance to the sink along the heaviest out-edge, reference abpoa_graph.c:233-274
        remain_.assign(n, 0);
        for (int i = 0; i < n; ++i) deg[i] = nodes_[i].out_id.size();
        q.clear(); q.push_back(SINK); head = 0; remain_[SINK] = -1;
        bool ok = false;
        while (head < q.size()) {
            const int cur = q[head++];
            const PoaNode &c = nodes_[cur];
            if (cur != SINK) {
                int max_w = -1, max_id = -1;
                for (int i = 0; i < c.out_id.size(); ++i) if (c.out_w[i] > max_w) { max_w = c.out_w[i]; max_id = c.out_id[i]; }
                remain_[cur] = remain_[max_id] + 1;
            }
            if (cur == SRC) { ok = true; break; }
            for (int i = 0; i < c.in_id.size(); ++i) if (--deg[c.in_id[i]] == 0) q.push_back(c.in_id[i]);
        }
        if (!ok) throw std::runtime_error("failed to set node remain");
    } else remain_.clear();
    sorted_ = true;
}

void PoaGraph::flatten(bool banded, FlatProblem *fp) const {
    const int n = (int)nodes_.size();
    fp->row_base.resize(n); fp->row_node_id.resize(n);
    fp->pred_off.resize(n + 1); fp->out_off.resize(n + 1);
    fp->pred_row.clear(); fp->out_row.clear();
    if (banded) { fp->row_remain.resize(n); fp->left.assign(n, n); fp->right.assign(n, 0); }   // reset :303-308
    else { fp->row_remain.clear(); fp->left.clear(); fp->right.clear(); }
    for (int r = 0; r < n; ++r) {
        const int id = index_to_node_[r]; const PoaNode &nd = nodes_[id];
        fp->row_base[r] = nd.base; fp->row_node_id[r] = id;
        if (banded) fp->row_remain[r] = remain_[id];
        fp->pred_off[r] = (int)fp->pred_row.size(); fp->out_off[r] = (int)fp->out_row.size();
        if (r > 0) for (int j = 0; j < nd.in_id.size(); ++j) fp->pred_row.push_back(node_to_index_[nd.in_id[j]]);
        for (int j = 0; j < nd.out_id.size(); ++j) fp->out_row.push_back(node_to_index_[nd.out_id[j]]);
    }
    fp->pred_off[n] = (int)fp->pred_row.size(); fp->out_off[n] = (int)fp->out_row.size();
    if (fp->pred_row.empty()) fp->pred_row.push_back(0);
    if (fp->out_row.empty()) fp->out_row.push_back(0);
}

void PoaGraph::flatten_into(bool with_remain, uint8_t *row_base, int32_t *row_node_id, int32_t *row_remain, int32_t *pred_off,
                            int32_t *pred_row, int32_t *out_off, int32_t *out_row) const {
    const int n = (int)nodes_.size();
    int np = 0, no = 0;
    for (int r = 0; r < n; ++r) {
        const int id = index_to_node_[r]; const PoaNode &nd = nodes_[id];
        row_base[r] = nd.base; row_node_id[r] = id; row_remain[r] = with_remain ? remain_[id] : 0;
        pred_off[r] = np; out_off[r] = no;
        for (int j = 0; j < nd.in_id.size(); ++j) pred_row[np++] = node_to_index_[nd.in_id[j]];
        for (int j = 0; j < nd.out_id.size(); ++j) out_row[no++] = node_to_index_[nd.out_id[j]];
    }
    pred_off[n] = np; out_off[n] = no;
}

void PoaGraph::consensus(std::vector<int> *node_ids, std::vector<uint8_t> *bases, std::vector<int> *cov) const {
    const int n = (int)nodes_.size();
    node_ids->clear(); bases->clear(); cov->clear();
    if (n <= 2) return;
    std::vector<int> deg(n), score(n, 0), max_out(n, -1), q; q.reserve(n);
    for (int i = 0; i < n; ++i) deg[i] = nodes_[i].out_id.size();
    q.push_back(SINK); size_t head = 0;
    while (head < q.size()) {                                          // reference abpoa_output.c:361-409
        const int cur = q[head++]; const PoaNode &c = nodes_[cur];
        if (cur == SINK) { max_out[cur] = -1; score[cur] = 0; }
        else if (cur == SRC) {
            int path_score = -1, path_max_w = -1, max_id = -1;
            for (int i = 0; i < c.out_id.size(); ++i) {
                const int o = c.out_id[i], ow = c.out_w[i];
                if (ow > path_max_w || (ow == path_max_w && score[o] > path_score)) { max_id = o; path_score = score[o]; path_max_w = ow; }
            }
            max_out[cur] = max_id; break;
        } else {
            int max_w = INT_MIN, max_id = -1;
            for (int i = 0; i < c.out_id.size(); ++i) {
                const int o = c.out_id[i], ow = c.out_w[i];
                if (max_w < ow) { max_w = ow; max_id = o; }
                else if (max_w == ow && score[max_id] <= score[o]) max_id = o;
            }
            score[cur] = max_w + score[max_id]; max_out[cur] = max_id;
        }
        for (int i = 0; i < c.in_id.size(); ++i) if (--deg[c.in_id[i]] == 0) q.push_back(c.in_id[i]);
    }
    for (int cur = max_out[SRC]; cur != SINK && cur >= 0; cur = max_out[cur]) {   // :343-356
        node_ids->push_back(cur); bases->push_back(nodes_[cur].base); cov->push_back(nodes_[cur].n_read);
    }
}

void PoaGraph::rc_msa(int m, int *msa_len, std::vector<std::vector<uint8_t>> *rows, std::vector<int> *node_col) const {
    const int n = (int)nodes_.size();
    rows->clear(); *msa_len = 0; node_col->assign(n, -1);
    if (n <= 2 || !use_read_ids_) return;
    // MSA rank: depth-first variant of the Kahn walk, one rank per aligned group, reference abpoa_graph.c:315-375
    std::vector<int> deg(n), rank(n, 0), st; st.reserve(n);
    for (int i = 0; i < n; ++i) deg[i] = nodes_[i].in_id.size();
    int msa_rank = 0; st.push_back(SRC); rank[SRC] = -1;
    bool done = false;
    while (!st.empty()) {
        const int cur = st.back(); st.pop_back();
        const PoaNode &c = nodes_[cur];
        if (rank[cur] < 0) {
            rank[cur] = msa_rank;
            for (int i = 0; i < c.aligned.size(); ++i) rank[c.aligned[i]] = msa_rank;
            ++msa_rank;
        }
        if (cur == SINK) { done = true; break; }
        for (int i = 0; i < c.out_id.size(); ++i) {
            const int o = c.out_id[i];
            if (--deg[o] != 0) continue;
            const PoaNode &on = nodes_[o];
            bool ready = true;
            for (int j = 0; j < on.aligned.size(); ++j) if (deg[on.aligned[j]] != 0) { ready = false; break; }
            if (!ready) continue;
            st.push_back(o); rank[o] = -1;
            for (int j = 0; j < on.aligned.size(); ++j) { st.push_back(on.aligned[j]); rank[on.aligned[j]] = -1; }
        }
    }
    if (!done) throw std::runtime_error("error in set_msa_rank");
    *msa_len = rank[SINK] - 1;
    rows->assign(tot_reads_, std::vector<uint8_t>((size_t)(*msa_len > 0 ? *msa_len : 0), (uint8_t)m));
    for (int i = 2; i < n; ++i) {                                      // reference abpoa_output.c:141-150, :103-120
        const PoaNode &nd = nodes_[i];
        int rk = rank[i];
        for (int j = 0; j < nd.aligned.size(); ++j) if (rank[nd.aligned[j]] > rk) rk = rank[nd.aligned[j]];
        (*node_col)[i] = rk - 1;
        const std::vector<uint64_t> &ids = read_ids_[i];
        for (int e = 0; e < nd.out_id.size(); ++e) for (int wd = 0; wd < words_; ++wd) {
            uint64_t num = ids[(size_t)e * words_ + wd];
            while (num) {
                const int b = __builtin_ctzll(num);
                const int read = wd * 64 + b;
                if (read < tot_reads_) (*rows)[read][rk - 1] = nd.base;
                num &= num - 1;
            }
        }
    }
}

}  // namespace abpoa_hip
