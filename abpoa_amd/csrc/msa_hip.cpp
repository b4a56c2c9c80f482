// The product's binding of the read-set driver to the HIP engine: every group gets its own BatchStream.
#include <memory>
#include <mutex>
#include "batch_stream.h"
#include "msa_batch.h"

namespace abpoa_hip {
namespace {
class HipGroupAligner : public GroupAligner {
  public:
    int init() { return bs_.open(engine_device()); }
    ~HipGroupAligner() override { add_global_stats(bs_.take_stats()); bs_.close(); }
    int prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *shapes) override { return bs_.prepare(sc, n, shapes, BS_FRESH_BAND); }
    ProblemSlots slots(int i) override { return bs_.slots(i); }
    int run() override { return bs_.run(); }
    int status(int i) override { return bs_.rec(i).status; }
    int64_t n_cells(int i) override { return bs_.rec(i).n_cells; }
    int n_cigar(int i) override { return bs_.rec(i).n_cigar; }
    const uint64_t *cigar(int i) override { return bs_.cigar(i); }
  private:
    BatchStream bs_;
};
GroupAligner *make_hip_aligner() {
    std::unique_ptr<HipGroupAligner> a(new HipGroupAligner());
    if (a->init() != 0) return nullptr;
    return a.release();
}
abpoa_hip_msa_timing_t g_timing;
}  // namespace
}  // namespace abpoa_hip

extern "C" {
int abpoa_hip_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets,
                        abpoa_hip_msa_t *out, unsigned flags, int n_threads) {
    if (abpoa_hip::engine_device() < 0) { int rc = abpoa_hip_init(0); if (rc) return rc; }
    return abpoa_hip::run_msa_batch(sc, n_sets, sets, out, flags, n_threads, 0, abpoa_hip::make_hip_aligner, &abpoa_hip::g_timing);
}
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out) { *out = abpoa_hip::g_timing; }
}
