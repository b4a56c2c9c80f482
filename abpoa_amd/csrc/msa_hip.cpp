// The product's binding of the read-set driver to the HIP engine: every group gets its own BatchStream.
#include <algorithm>
#include <memory>
#include <thread>
#include <mutex>
#include "batch_stream.h"
#include "msa_batch.h"
#include "msa_device.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

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
    int best_score(int i) override { return bs_.rec(i).best_score; }
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
    using namespace abpoa_hip;
    if (engine_device() < 0) { int rc = abpoa_hip_init(0); if (rc) return rc; }
    bool plain = true;      // per-base weights and the strand retry are host-driver features
    for (int s = 0; plain && sets && s < n_sets; ++s) plain = sets[s].weights == nullptr;
    if (n_sets > 0 && sc && sets && out && plain && msa_device_eligible(sc, flags)) {
        // device-resident driver first; sets that outgrow a device capacity (and whole jobs that do not fit) go to the host driver
        if (n_threads <= 0) n_threads = effective_host_cores();
        for (int s = 0; s < n_sets; ++s) { if (sets[s].n_reads < 0) return ABPOA_HIP_EINVAL; for (int r = 0; r < sets[s].n_reads; ++r) if (sets[s].lens[r] <= 0 || !sets[s].seqs[r]) return ABPOA_HIP_EINVAL; }
        // Pass 1 gives every set 3x its longest read in graph-node slots (5 %-error reads need ~2.5x), pass 2 retries the sets that
        // outgrew that with 6x, whatever is left goes to the host driver.  A pass that does not fit the device memory is split in halves.
        for (int s = 0; s < n_sets; ++s) memset(&out[s], 0, sizeof(out[s]));
        std::vector<int> todo(n_sets), left; for (int s = 0; s < n_sets; ++s) todo[s] = s;
        DeviceRunStats tot; memset(&tot, 0, sizeof(tot));
        bool device_ok = true;
        const double factors[2] = {3.0, 6.0};
        for (int pass = 0; pass < 2 && device_ok && !todo.empty(); ++pass) {
            left.clear();
            size_t chunk = todo.size();
            for (size_t at = 0; at < todo.size() && device_ok;) {
                const size_t nb = std::min(chunk, todo.size() - at);
                std::vector<abpoa_hip_readset_t> sub(nb); std::vector<abpoa_hip_msa_t> sub_out(nb);
                for (size_t i = 0; i < nb; ++i) sub[i] = sets[todo[at + i]];
                std::vector<int> fb; DeviceRunStats ds;
                const int rc = run_msa_device(sc, (int)nb, sub.data(), sub_out.data(), n_threads, &fb, &ds, factors[pass]);
                if (rc == ABPOA_HIP_ENOMEM && nb > 1) { chunk = (nb + 1) / 2; continue; }           // split and retry this chunk
                if (rc != ABPOA_HIP_OK) { if (rc != ABPOA_HIP_ENOMEM && rc != ABPOA_HIP_EINVAL) return rc; device_ok = false; break; }
                for (size_t i = 0; i < nb; ++i) out[todo[at + i]] = sub_out[i];
                for (int f : fb) left.push_back(todo[at + f]);
                tot.prepare_ms += ds.prepare_ms; tot.rows_ms += ds.rows_ms; tot.tail_ms += ds.tail_ms; tot.fuse_ms += ds.fuse_ms; tot.device_s += ds.device_s; tot.cons_s += ds.cons_s;
                tot.total_s += ds.total_s; tot.n_cells += ds.n_cells; tot.algo_bytes += ds.algo_bytes; tot.n_alignments += ds.n_alignments; tot.n_rounds += ds.n_rounds;
                if (getenv("ABPOA_HIP_VERBOSE")) fprintf(stderr, "[abpoa-hip] device-resident driver (pass %d, node slots %.0fx): %zu sets, %d rounds: prepare %.1f ms, dp rows %.1f ms, backtrack %.1f ms, fuse %.1f ms; device wall %.1f ms, results %.1f ms, total %.1f ms; %zu sets outgrew a device capacity\n",
                                                         pass + 1, factors[pass], nb, ds.n_rounds, ds.prepare_ms, ds.rows_ms, ds.tail_ms, ds.fuse_ms, ds.device_s * 1e3, ds.cons_s * 1e3, ds.total_s * 1e3, fb.size());
                at += nb;
            }
            todo.swap(left);
        }
        if (device_ok) {
            StreamStats ss; ss.n_launches = tot.n_rounds; ss.n_alignments = tot.n_alignments; ss.n_cells = tot.n_cells; ss.algo_bytes = tot.algo_bytes;
            ss.kernel_ms = tot.rows_ms; ss.tail_ms = tot.tail_ms; add_global_stats(ss);
            memset(&g_timing, 0, sizeof(g_timing));
            g_timing.engine_s = tot.device_s; g_timing.cons_s = tot.cons_s; g_timing.total_s = tot.total_s; g_timing.n_rounds = tot.n_rounds; g_timing.n_threads = n_threads; g_timing.n_groups = 1;
            g_timing.host_sort_s = tot.prepare_ms / 1e3; g_timing.host_fuse_s = tot.fuse_ms / 1e3;      // device kernels now: graph -> rows, cigar -> graph
            g_timing.pad = (int32_t)todo.size();             // how many sets take the host driver
            if (todo.empty()) return ABPOA_HIP_OK;
            std::vector<abpoa_hip_readset_t> sub(todo.size()); std::vector<abpoa_hip_msa_t> sub_out(todo.size());
            for (size_t i = 0; i < todo.size(); ++i) sub[i] = sets[todo[i]];
            abpoa_hip_msa_timing_t t2;
            const int rc2 = run_msa_batch(sc, (int)todo.size(), sub.data(), sub_out.data(), flags, n_threads, 0, make_hip_aligner, &t2);
            if (rc2 != ABPOA_HIP_OK) { for (int s = 0; s < n_sets; ++s) abpoa_hip_free_msa(&out[s]); return rc2; }
            for (size_t i = 0; i < todo.size(); ++i) out[todo[i]] = sub_out[i];
            return ABPOA_HIP_OK;
        }
        for (int s = 0; s < n_sets; ++s) abpoa_hip_free_msa(&out[s]);
    }
    return run_msa_batch(sc, n_sets, sets, out, flags, n_threads, 0, make_hip_aligner, &g_timing);
}
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out) { *out = abpoa_hip::g_timing; }
}
