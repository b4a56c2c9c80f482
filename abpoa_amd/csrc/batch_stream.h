// BatchStream: one HIP stream + its pinned staging / HBM pools.  A batch of alignments is
//   prepare(shapes)  -> layout computed, pools grown, host slots handed out (pinned memory, zero copy)
//   fill slots       -> caller writes each problem straight into its slot (any thread)
//   run()            -> one H2D copy, one DP-kernel launch, one D2H copy, synchronise; overflowed arenas are retried
//   rec()/cigar()    -> results read in place from pinned memory
// Several BatchStreams may be driven concurrently from different host threads (one per read-set group);
// the flat C API (abpoa_hip_align_batch) uses a process-wide default instance under a mutex.
#pragma once
#include <stdint.h>
#include <vector>
#include <hip/hip_runtime.h>
#include "engine.h"
#include "batch_types.h"
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

enum : unsigned {
    BS_TRACE = 0x1,             // keep per-row arg-max, allow fetch_trace()
    BS_FRESH_BAND = 0x2,        // max_pos_left/right start as (n_rows, 0) for every row: initialised on the device
    BS_WANT_BAND_STATE = 0x4,   // copy max_pos_left/right back to the host
};

struct Blob {                   // device buffer with optional pinned host mirror; grow-only
    uint8_t *dev = nullptr, *host = nullptr; size_t cap = 0; bool mirrored;
    explicit Blob(bool m) : mirrored(m) {}
    int reserve(size_t n);
    void release();
};

struct StreamStats { int64_t n_launches = 0, n_alignments = 0, n_cells = 0, algo_bytes = 0, n_need_scores = 0; double kernel_ms = 0, h2d_ms = 0, d2h_ms = 0, tail_ms = 0, rounds_ms = 0;
        int64_t rounds_launches = 0, rounds_algo_bytes = 0; };

class BatchStream {
  public:
    BatchStream() : in_(true), out_(true), planes_(false) {}
    ~BatchStream() { close(); }
    int open(int device);
    void close();
    int prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *shapes, unsigned flags);
    ProblemSlots slots(int i) const;
    int run();
    int n() const { return n_; }
    const AlnOut &rec(int i) const { return recs_[i]; }
    const AlnDesc &desc(int i) const { return desc_[i]; }
    const uint64_t *cigar(int i) const;
    const int32_t *left(int i) const;
    const int32_t *right(int i) const;
    int fetch_trace(int i, const uint8_t *row_active, abpoa_hip_trace_t *T);   // after run(), BS_TRACE only
    StreamStats take_stats() { StreamStats s = stats_; stats_ = StreamStats(); return s; }

  private:
    bool open_ = false; int device_ = -1;
    hipStream_t stream_ = nullptr; hipEvent_t ev_[5] = {};
    Blob in_, out_, planes_;
    abpoa_hip_scoring_t sc_{}; std::vector<int32_t> mat_;
    unsigned flags_ = 0; int n_ = 0, P_ = 1;
    // arena capacities (cells): score records full width / estimate, direction-plane arenas likewise
    std::vector<AlnDesc> desc_; std::vector<AlnOut> recs_; std::vector<int64_t> full_cells_, est_cells_, dir_full_cells_, dir_est_cells_;
    std::vector<std::vector<uint8_t>> trace_arena_;     // BS_TRACE: arena of every finished alignment, copied out before a retry pass re-uses the device arenas
    int64_t rows_tot_ = 0, preds_tot_ = 0, outs_tot_ = 0, q_tot_ = 0, cig_tot_ = 0;
    size_t o_desc_ = 0, o_mat_ = 0, o_query_ = 0, o_base_ = 0, o_sdist_ = 0, o_pd_ = 0, o_nid_ = 0, o_rem_ = 0, o_act_ = 0, o_poff_ = 0, o_pred_ = 0, o_ooff_ = 0, o_out_ = 0, in_bytes_ = 0;
    size_t o_rec_ = 0, o_left_ = 0, o_right_ = 0, o_bsn_ = 0, o_esn_ = 0, o_coff_ = 0, o_rmi_ = 0, o_cig_ = 0, out_bytes_ = 0;
    StreamStats stats_;
};

void set_err(const char *fmt, ...);
const char *thread_last_error();      // the last message set_err formatted on the calling thread ("" if none since clear_thread_error)
void clear_thread_error();
void make_lds_plan(const abpoa_hip_scoring_t *sc, int max_qlen, int max_bits, int64_t est_cols, int n_aln, LdsPlan *L);
int engine_device();            // device the process is bound to, or -1
void add_global_stats(const StreamStats &s);

}  // namespace abpoa_hip
