// TEST INFRASTRUCTURE.  Builds the product's HOST sources (poa_graph.cpp, msa_batch.cpp) into a CPU-only library in
// which the batch aligner is backed by the oracle, so that graph fusion, row ordering, flattening into staging slots,
// the grouped lock-step driver, consensus and MSA can be checked against the reference's outputs without a GPU.
// Never shipped: the product library binds the driver to the HIP engine only (abpoa_amd/csrc/msa_hip.cpp).
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "../abpoa_amd/csrc/msa_batch.h"
extern "C" {
#include "../oracle/abpoa_dp_oracle.h"
}

namespace {
using namespace abpoa_hip;
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
            int rc = abpoa_oracle_align(&sc_, &pb, &res_[i], 0);
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
    return abpoa_hip::run_msa_batch(sc, n_sets, sets, out, flags, n_threads, n_sets >= 4 ? 2 : 1, make_oracle_aligner, &g_timing);
}
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out) { *out = g_timing; }
const char *abpoa_hip_last_error(void) { return "cpu shim"; }
}
