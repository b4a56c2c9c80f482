// Host-side partial-order graph for the read-set batch driver.
//
// The DP engine consumes a graph snapshot in topological order (include/abpoa_hip.h,
// abpoa_hip_problem_t) and returns a graph cigar; this class owns what happens between two
// alignments of one read-set: fusing the cigar into the graph, re-deriving the row order and the
// "remaining length" used by the adaptive band, and finally calling the consensus / MSA.
// Behaviour is pinned to abPOA v1.4.1 because consensus parity depends on node numbering, edge
// insertion order and integer tie-breaks:
//   fusion            src/abpoa_graph.c:596-672   (abpoa_add_subgraph_alignment)
//   edges             src/abpoa_graph.c:418-484   (abpoa_add_graph_edge)
//   row order         src/abpoa_graph.c:186-231   (abpoa_BFS_set_node_index)
//   remaining length  src/abpoa_graph.c:233-274   (abpoa_BFS_set_node_remain)
//   consensus         src/abpoa_output.c:361-415  (abpoa_heaviest_bundling), :343-356
//   MSA               src/abpoa_graph.c:315-375, src/abpoa_output.c:103-166
// The data layout is our own: structure-of-arrays nodes with small inline edge lists.
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

// Tiny vector with N inline slots (POA nodes have 1-3 edges almost always).
template <typename T, int N>
class SmallVec {
  public:
    SmallVec() : n_(0), cap_(N), heap_(nullptr) {}
    SmallVec(const SmallVec &o) : n_(0), cap_(N), heap_(nullptr) { *this = o; }
    SmallVec &operator=(const SmallVec &o) {
        if (this == &o) return *this;
        clear(); for (int i = 0; i < o.n_; ++i) push_back(o[i]); return *this;
    }
    ~SmallVec() { delete[] heap_; }
    int size() const { return n_; }
    void clear() { n_ = 0; }
    T &operator[](int i) { return heap_ ? heap_[i] : inl_[i]; }
    const T &operator[](int i) const { return heap_ ? heap_[i] : inl_[i]; }
    void push_back(const T &v) {
        if (n_ == cap_) {
            int nc = cap_ * 2; T *h = new T[nc];
            for (int i = 0; i < n_; ++i) h[i] = (*this)[i];
            delete[] heap_; heap_ = h; cap_ = nc;
        }
        (*this)[n_++] = v;
    }
  private:
    int n_, cap_; T inl_[N]; T *heap_;
};

struct PoaNode {
    SmallVec<int, 3> in_id;
    SmallVec<int, 3> out_id;
    SmallVec<int, 3> out_w;
    SmallVec<int, 2> aligned;     // nodes aligned to this one (mismatch alternatives share an MSA column)
    int n_read = 0;
    uint8_t base = 0;
};

// Flattened snapshot handed to the engine; owns its arrays.
struct FlatProblem {
    std::vector<uint8_t> row_base, row_active;
    std::vector<int32_t> row_node_id, row_remain, pred_off, pred_row, out_off, out_row, left, right;
    abpoa_hip_problem_t view(const uint8_t *query, int qlen);
};

class PoaGraph {
  public:
    static constexpr int SRC = 0, SINK = 1;      // reference ABPOA_SRC_NODE_ID / ABPOA_SINK_NODE_ID
    PoaGraph() { reset(0, false); }
    // tot_reads is needed up front when read ids are tracked (one bit per read per out-edge)
    void reset(int tot_reads, bool use_read_ids);
    int n_nodes() const { return (int)nodes_.size(); }
    bool empty() const { return nodes_.size() <= 2; }
    const PoaNode &node(int id) const { return nodes_[id]; }

    // Fuse an alignment (graph cigar from the DP) of read `read_id` into the graph; with an empty graph
    // the read becomes the backbone chain and the cigar is ignored.
    void add_alignment(const uint8_t *seq, int len, const uint64_t *cigar, int n_cigar, int read_id, const int32_t *weight = nullptr);      // weight: per-base edge weights (qv), NULL = 1

    // Row order + (if `with_remain`) heaviest-path remaining length.
    void topological_sort(bool with_remain);
    // Snapshot of the whole graph (row 0 = source, last row = sink) for the engine.
    void flatten(bool banded, FlatProblem *out) const;
    // Same snapshot written straight into the engine's pinned staging slots (sizes: n_nodes(), n_edges()).
    void flatten_into(bool with_remain, uint8_t *row_base, int32_t *row_node_id, int32_t *row_remain, int32_t *pred_off,
                      int32_t *pred_row, int32_t *out_off, int32_t *out_row) const;
    int n_edges() const { return n_edges_; }
    // Rebuild the node table from flat fixed-capacity arrays (the device-resident graph of poa_device.h); read ids are not carried.
    void import_nodes(int n, const uint8_t *base, const uint8_t *nin, const uint8_t *nout, const uint8_t *naln, const int32_t *in_id, int in_cap,
                      const int32_t *out_id, const int32_t *out_w, int out_cap, const int32_t *aligned, int aln_cap, const int32_t *n_read);

    // Single heaviest-bundling consensus: node ids of the path, bases and per-base coverage.
    void consensus(std::vector<int> *node_ids, std::vector<uint8_t> *bases, std::vector<int> *cov) const;
    // Row-column MSA (reads only): n_reads rows of msa_len codes, gap = m.  Requires use_read_ids.
    void rc_msa(int m, int *msa_len, std::vector<std::vector<uint8_t>> *rows, std::vector<int> *node_col) const;

    const std::vector<int> &index_to_node() const { return index_to_node_; }
    const std::vector<int> &node_to_index() const { return node_to_index_; }
    const std::vector<int> &remain() const { return remain_; }

  private:
    int add_node(uint8_t base);
    void add_edge(int from, int to, bool check_edge, int w, bool add_read_id, int read_id);
    int aligned_with_base(int node_id, uint8_t base) const;
    void add_aligned(int node_id, int new_id);

    std::vector<PoaNode> nodes_;
    std::vector<std::vector<uint64_t>> read_ids_;   // per node: out_edge * words_ + w   (only if use_read_ids_)
    std::vector<int> index_to_node_, node_to_index_, remain_;
    mutable std::vector<int> scratch_deg_, scratch_q_;
    int tot_reads_ = 0, words_ = 0, n_edges_ = 0;
    bool use_read_ids_ = false, sorted_ = false;
};

}  // namespace abpoa_hip
