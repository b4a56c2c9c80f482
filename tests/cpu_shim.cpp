// TEST INFRASTRUCTURE.  Builds the product's HOST sources (poa_graph.cpp, msa_batch.cpp) into a CPU-only library in
// which the batch aligner is backed by the oracle, so that graph fusion, row ordering, flattening into staging slots,
// the grouped lock-step driver, consensus and MSA can be checked against the reference's outputs without a GPU.
// Never shipped: the product library binds the driver to the HIP engine only (abpoa_amd/csrc/msa_hip.cpp).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../abpoa_amd/csrc/msa_batch.h"
#include "../abpoa_amd/csrc/engine_options.h"
extern "C" {
#include "../oracle/abpoa_dp_oracle.h"
}

#include <atomic>
#include <stdio.h>
namespace {
using namespace abpoa_hip;
std::atomic<long long> n_dir_checked{0}, n_dir_steps{0}, n_dir_ambig{0}, n_dir_lit{0};
class OracleGroupAligner : public GroupAligner {
  public:
    ~OracleGroupAligner() override { clear(); }
    const int32_t *left(int i) override { return p_[i].left.data(); }
    const int32_t *right(int i) override { return p_[i].right.data(); }
    int prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *sh, int band) override {
        (void)band;      // (the oracle updates max_pos_left/right in place: a seeded start is whatever the caller writes into the slots after this)
        clear(); sc_ = *sc; n_ = n; p_.resize(n); res_.assign(n, abpoa_hip_result_t());
        for (int i = 0; i < n; ++i) {
            P &p = p_[i]; const int gn = sh[i].n_rows;
            p.query.assign(sh[i].qlen + 1, 0); p.base.assign(gn, 0); p.active.assign(gn, 1); p.nid.assign(gn, 0); p.remain.assign(gn, 0);
            p.poff.assign(gn + 1, 0); p.ooff.assign(gn + 1, 0); p.pred.assign(sh[i].n_pred + 1, 0); p.out.assign(sh[i].n_out + 1, 0);
            p.left.assign(gn, gn); p.right.assign(gn, 0);           // fresh band state, reference abpoa_graph.c:303-308
            p.gn = gn; p.qlen = sh[i].qlen;
        }
        return 0;
    }
    ProblemSlots slots(int i) override {
        P &p = p_[i]; ProblemSlots s;
        s.query = p.query.data(); s.row_base = p.base.data(); s.row_node_id = p.nid.data(); s.row_remain = p.remain.data(); s.row_active = p.active.data();
        s.pred_off = p.poff.data(); s.pred_row = p.pred.data(); s.out_off = p.ooff.data(); s.out_row = p.out.data(); s.left = p.left.data(); s.right = p.right.data();
        return s;
    }
    int run() override {
        for (int i = 0; i < n_; ++i) {
            P &p = p_[i]; abpoa_hip_problem_t pb;
            pb.n_rows = p.gn; pb.qlen = p.qlen; pb.query = p.query.data(); pb.row_base = p.base.data(); pb.row_node_id = p.nid.data();
            pb.row_remain = p.remain.data(); pb.row_active = p.active.data(); pb.pred_off = p.poff.data(); pb.pred_row = p.pred.data();
            pb.out_off = p.ooff.data(); pb.out_row = p.out.data(); pb.max_pos_left = p.left.data(); pb.max_pos_right = p.right.data();
            // ABPOA_SHIM_DIR_CHECK=1: keep the trace and let oracle/dir_model.c build and walk the direction plane of this alignment;
            // its cigar and result fields must equal the value-comparing backtrack's (tests/test_dir_model.py)
            const bool dir_check = getenv("ABPOA_SHIM_DIR_CHECK") && atoi(getenv("ABPOA_SHIM_DIR_CHECK"));      // (read per call: tests switch it on and off)
            abpoa_oracle_trace_t tr; memset(&tr, 0, sizeof(tr));
            int rc = abpoa_oracle_align(&sc_, &pb, &res_[i], dir_check ? &tr : 0);
            if (dir_check && rc == 0 && res_[i].status == 0) {
                abpoa_hip_result_t r2; int64_t st[10];
                const int rc2 = abpoa_oracle_dir_walk(&sc_, &pb, &tr, res_[i].best_row, res_[i].best_col, &r2, st);
                if (rc2 == 0) {
                    const abpoa_hip_result_t &a = res_[i];
                    bool same = a.n_cigar == r2.n_cigar && a.node_s == r2.node_s && a.node_e == r2.node_e && a.query_s == r2.query_s && a.query_e == r2.query_e &&
                                a.n_aln_bases == r2.n_aln_bases && a.n_matched_bases == r2.n_matched_bases && st[2] == 0 && st[3] == 0 && st[7] == 0;
                    for (int k = 0; same && k < a.n_cigar; ++k) same = a.cigar[k] == r2.cigar[k];
                    free(r2.cigar);
                    if (!same) { fprintf(stderr, "[cpu shim] direction-plane model differs from the oracle backtrack (rows %d, qlen %d; derived-F mismatches %lld, uE mismatches %lld)\n", p.gn, p.qlen, (long long)st[2], (long long)st[3]); abpoa_oracle_free_trace(&tr); return ABPOA_HIP_EBACKTRACK; }
                    n_dir_checked.fetch_add(1); n_dir_steps.fetch_add(st[4]); n_dir_ambig.fetch_add(st[8]); n_dir_lit.fetch_add(st[5]);
                } else if (rc2 != ABPOA_HIP_EINVAL) { abpoa_oracle_free_trace(&tr); return rc2; }
            }
            if (dir_check) abpoa_oracle_free_trace(&tr);
            if (rc && res_[i].status == 0) return rc;
        }
        return 0;
    }
    int status(int i) override { return res_[i].status; }
    int64_t n_cells(int i) override { return res_[i].n_cells; }
    int best_score(int i) override { return res_[i].best_score; }
    int n_cigar(int i) override { return res_[i].n_cigar; }
    const uint64_t *cigar(int i) override { return res_[i].cigar; }
  private:
    struct P { std::vector<uint8_t> query, base, active; std::vector<int32_t> nid, remain, poff, ooff, pred, out, left, right; int gn, qlen; };
    void clear() { for (auto &r : res_) free(r.cigar); res_.clear(); }
    abpoa_hip_scoring_t sc_; int n_ = 0; std::vector<P> p_; std::vector<abpoa_hip_result_t> res_;
};
GroupAligner *make_oracle_aligner() { return new OracleGroupAligner(); }
abpoa_hip_msa_timing_t g_timing;
}  // namespace

extern "C" {
int abpoa_hip_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, unsigned flags, int n_threads) {
    abpoa_hip::refresh_options();
    return abpoa_hip::run_msa_batch(sc, n_sets, sets, out, flags, n_threads, n_sets >= 4 ? 2 : 1, make_oracle_aligner, &g_timing);
}
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out) { *out = g_timing; }
void abpoa_shim_dir_counts(long long *o) { o[0] = n_dir_steps.exchange(0); o[1] = n_dir_ambig.exchange(0); o[2] = n_dir_lit.exchange(0); }
long long abpoa_shim_dir_checked(void) { return n_dir_checked.exchange(0); }      // alignments the direction-plane model confirmed since the last call
const char *abpoa_hip_last_error(void) { return "cpu shim"; }
// (life-cycle entries so that host programs written against include/abpoa_hip.h -- abpoa_amd/host/abpoa_batch.c -- link against this test build too)
int abpoa_hip_init(int) { return 0; }
void abpoa_hip_shutdown(void) {}
}
