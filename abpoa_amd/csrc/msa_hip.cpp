// The product's binding of the read-set driver to the HIP engine: every group gets its own BatchStream.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <memory>
#include <thread>
#include <mutex>
#include "engine_options.h"
#include "batch_stream.h"
#include "msa_batch.h"
#include "msa_device.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <map>
#include <vector>

namespace abpoa_hip {
namespace {
class HipGroupAligner : public GroupAligner {
  public:
    int init() { return bs_.open(engine_device()); }
    ~HipGroupAligner() override { add_global_stats(bs_.take_stats()); bs_.close(); }
    int prepare(const abpoa_hip_scoring_t *sc, int n, const BatchShape *shapes, int band) override {
        return bs_.prepare(sc, n, shapes, band == GA_BAND_FRESH ? BS_FRESH_BAND : (band == GA_BAND_KEEP ? (BS_FRESH_BAND | BS_WANT_BAND_STATE) : BS_WANT_BAND_STATE));
    }
    const int32_t *left(int i) override { return bs_.left(i); }
    const int32_t *right(int i) override { return bs_.right(i); }
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

namespace abpoa_hip {
namespace {
bool env_on(const char *name) { const char *e = opt_env(name); return e && atoi(e) != 0; }
int env_int(const char *name, int dflt) { const char *e = opt_env(name); return e ? atoi(e) : dflt; }
bool strict_mode() { return env_on("ABPOA_HIP_STRICT"); }
void free_all(abpoa_hip_msa_t *out, int n) { for (int s = 0; s < n; ++s) abpoa_hip_free_msa(&out[s]); }

void add_stats(DeviceRunStats &t, const DeviceRunStats &d) {
    t.prepare_ms += d.prepare_ms; t.rows_ms += d.rows_ms; t.tail_ms += d.tail_ms; t.fuse_ms += d.fuse_ms;
    t.device_s += d.device_s; t.cons_s += d.cons_s; t.total_s += d.total_s;
    t.n_cells += d.n_cells; t.algo_bytes += d.algo_bytes; t.n_alignments += d.n_alignments; t.n_rounds += d.n_rounds;
    t.rounds_ms += d.rounds_ms; t.rounds_launches += d.rounds_launches; t.rounds_algo_bytes += d.rounds_algo_bytes;
}

// What the process learned about jobs of one shape (longest read by power of two, reads per set): when most sets of the last such job
// outgrew the 3x pass and a later pass ran in one piece, the next one starts there (noisy long reads: 50 x 10 kb at 15 % error grow to
// 3.9x; the doomed first pass is ~3 % of such a job).  Results do not depend on it.  ABPOA_HIP_NO_PASS_HINT=1: always start at 3x.
std::mutex g_hint_mu;
std::map<int, int> g_hint;
int job_shape_key(const abpoa_hip_readset_t *sets, const std::vector<int> &idx) {
    int mx = 1, nr = 0;
    for (int i : idx) {
        nr = std::max(nr, sets[i].n_reads);
        for (int r = 0; r < sets[i].n_reads; ++r) mx = std::max(mx, sets[i].lens[r]);
    }
    int lg = 0;
    while ((1 << lg) < mx) ++lg;
    return lg * 1024 + std::min(nr, 1023);
}

// The device passes over the sets `idx` on ONE device queue (device, slot): pass 1 gives every set 3x its longest read in graph-node
// slots (5 %-error reads need ~2.5x), the next ones retry the sets that outgrew that with 4.5x and 6x; whatever is left (listed in
// `left`) goes to the host driver.  A pass that does not fit the device memory is split in halves.
struct PassOut { int rc = ABPOA_HIP_OK; bool device_ok = true; DeviceRunStats tot; std::vector<int> left; std::map<int, int> why; };      // why: set -> reason it left its last pass
// reasons of the read-sets that the last batch call handed to the host driver (abpoa_hip_get_host_reasons; index = msa_device.h reason code, 11 = the job's
// options / a pass that did not fit)
std::mutex g_reason_mu;
int32_t g_host_reasons[MSA_HOST_REASONS];
PassOut device_passes(const abpoa_hip_scoring_t *sc, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, const std::vector<int> &idx,
                      int n_threads, int device, int slot, unsigned flags) {
    PassOut R;
    memset(&R.tot, 0, sizeof(R.tot));
    std::vector<int> todo = idx, left;
    // (the last pass bounds nothing: a set's graph cannot have more nodes than its reads have bases, and run_msa_device takes the smaller of the two -- so
    //  node slots are never what sends a set to the host driver; round-4 fuzzing: 22 of 813 sets, all for that reason -- protein sets at 15 % error)
    const double factors[4] = {3.0, 4.5, 6.0, 4096.0};
    constexpr int NPASS = 4;
    const bool verbose = opt_env("ABPOA_HIP_VERBOSE") != nullptr;
    const int key = job_shape_key(sets, idx);
    int first_pass = 0;
    {   // (profiling runs of one step: start where a warmed-up process would)
        const int fp = env_int("ABPOA_HIP_FIRST_PASS", 0);
        if (fp >= 1 && fp < NPASS) first_pass = fp;
    }
    if (!env_on("ABPOA_HIP_NO_PASS_HINT")) {
        std::lock_guard<std::mutex> lk(g_hint_mu);
        auto it = g_hint.find(key);
        if (it != g_hint.end()) first_pass = it->second;
    }
    bool most_outgrew = false;
    // sets with a node out of edge slots (run_msa_device marks them): more node slots do not help, the LAST pass does -- it has an edge slot per read at
    // every node (msa_device.cpp `roomy`) -- so they skip the passes in between; what the last pass marks is the caller's
    std::vector<int> deferred, hopeless;
    int n_small = 0, n_done = 0;      // (sets that would also have fitted the 3x estimate / sets that finished)
    for (int pass = first_pass; pass < NPASS && R.device_ok && (!todo.empty() || !deferred.empty()); ++pass) {
        if (todo.empty()) pass = NPASS - 1;
        if (pass == NPASS - 1) { todo.insert(todo.end(), deferred.begin(), deferred.end()); deferred.clear(); std::sort(todo.begin(), todo.end()); }
        left.clear();
        size_t chunk = todo.size();
        bool halved = false;          // (the pass did not fit the device memory in the pieces first tried)
        {   // wide-band jobs: passes of what the device holds at once (msa_device.h)
            std::vector<abpoa_hip_readset_t> all_(todo.size());
            for (size_t i = 0; i < todo.size(); ++i) all_[i] = sets[todo[i]];
            const int res_ = msa_device_resident_sets(sc, (int)all_.size(), all_.data());
            if (res_ > 0 && chunk > (size_t)res_) chunk = (size_t)res_;
            const int ps_ = env_int("ABPOA_HIP_PASS_SETS", 0);      // (tests: several passes on a small job)
            if (ps_ > 0 && chunk > (size_t)ps_) chunk = (size_t)ps_;
        }
        for (size_t at = 0; at < todo.size() && R.device_ok;) {
            const size_t nb = std::min(chunk, todo.size() - at);
            std::vector<abpoa_hip_readset_t> sub(nb);
            std::vector<abpoa_hip_msa_t> sub_out(nb);
            for (size_t i = 0; i < nb; ++i) sub[i] = sets[todo[at + i]];
            std::vector<int> fb, fb_why;
            DeviceRunStats ds;
            const int rc = run_msa_device(sc, (int)nb, sub.data(), sub_out.data(), n_threads, &fb, &ds, factors[pass], flags, device, slot, &fb_why);
            if (rc == ABPOA_HIP_ENOMEM && nb > 1) { chunk = (nb + 1) / 2; halved = true; continue; }      // split and retry this chunk
            if (rc != ABPOA_HIP_OK) {
                if (rc != ABPOA_HIP_ENOMEM && rc != ABPOA_HIP_EINVAL) { R.rc = rc; return R; }
                // not a job for the device path (does not fit even alone / shape): what is still open -- the leftovers of the chunks
                // already done in this pass and everything from here on -- goes back to the caller; finished results stay in out[]
                R.device_ok = false;
                for (size_t i = at; i < todo.size(); ++i) left.push_back(todo[i]);
                todo.swap(left);
                break;
            }
            for (size_t i = 0; i < nb; ++i) out[todo[at + i]] = sub_out[i];
            for (int f : fb) { if (f >= 0) left.push_back(todo[at + f]); else (pass < NPASS - 1 ? deferred : hopeless).push_back(todo[at + (-f - 1)]); }      // (f < 0: a full edge list)
            for (size_t i = 0; i < fb.size() && i < fb_why.size(); ++i) R.why[todo[at + (fb[i] < 0 ? -fb[i] - 1 : fb[i])]] = fb_why[i];
            add_stats(R.tot, ds);
            n_small += ds.n_fit_3x;
            n_done += (int)nb - (int)fb.size();
            if (verbose)
                fprintf(stderr, "[abpoa-hip] device-resident driver (device %d, pass %d, node slots %gx): %zu sets, %d rounds: prepare %.1f ms, "
                                "dp rows %.1f ms, backtrack %.1f ms, fuse %.1f ms; device wall %.1f ms, results %.1f ms, total %.1f ms; "
                                "%zu sets outgrew a device capacity\n",
                        device, pass + 1, factors[pass], nb, ds.n_rounds, ds.prepare_ms, ds.rows_ms, ds.tail_ms, ds.fuse_ms, ds.device_s * 1e3,
                        ds.cons_s * 1e3, ds.total_s * 1e3, fb.size());
            if (verbose && ds.rounds_launches)
                fprintf(stderr, "[abpoa-hip]   all-rounds kernel: %.1f ms (the phase times above are its duration split by the sets' clock ticks); "
                                "mean set busy %.0f %% of it; mean set, 10^6 ticks: prepare %.1f, row loop %.1f, backtrack %.1f, fuse %.1f\n",
                        ds.rounds_ms, 100.0 * ds.rounds_mean_over_max, ds.rounds_mticks[0], ds.rounds_mticks[1], ds.rounds_mticks[2], ds.rounds_mticks[3]);
            at += nb;
        }
        // most sets of the previous pass outgrew it and this one held most of them (in the pieces first tried): jobs of this shape start here next time
        const bool outgrew_now = left.size() * 2 >= todo.size();
        if (R.device_ok && pass > 0 && pass < NPASS - 1 && most_outgrew && !outgrew_now && !halved) { std::lock_guard<std::mutex> lk(g_hint_mu); g_hint[key] = pass; }
        if (R.device_ok) most_outgrew = outgrew_now;
        // the hint is dropped again when a job that started higher because of it turns out to fit 3x (a cleaner job of the same shape): more
        // graph and arena memory for nothing otherwise, for as long as the process lives
        if (R.device_ok && pass == first_pass && first_pass > 0 && n_done > 0 && n_small * 2 > n_done) {
            std::lock_guard<std::mutex> lk(g_hint_mu);
            g_hint.erase(key);
        }
        if (R.device_ok) todo.swap(left);
    }
    R.left = todo;
    R.left.insert(R.left.end(), deferred.begin(), deferred.end());      // (only when the device path gave up on the job)
    R.left.insert(R.left.end(), hopeless.begin(), hopeless.end());
    return R;
}

// ABPOA_GPU_DEVICES: "all", or a comma list of device ordinals (a repeated ordinal = two queues on that
// device); unset = the device the engine was initialised on.
std::vector<int> device_list() {
    std::vector<int> d;
    const char *e = opt_env("ABPOA_GPU_DEVICES");
    int n = 0;
    (void)hipGetDeviceCount(&n);
    if (e && *e) {
        if (!strcmp(e, "all")) { for (int i = 0; i < n; ++i) d.push_back(i); }
        else for (const char *q = e; *q;) {
            char *end;
            const long v = strtol(q, &end, 10);
            if (end == q) break;
            if (v >= 0 && v < n) d.push_back((int)v);
            if (*end && *end != ',') break;
            q = *end == ',' ? end + 1 : end;
        }
    }
    if (d.empty()) d.push_back(engine_device());
    if ((int)d.size() > MSA_DEVICE_SLOTS) d.resize(MSA_DEVICE_SLOTS);
    return d;
}

// Batches for the device queues: sets sorted by estimated DP cost (sum of read lengths x reads), heaviest first, dealt round-robin so that
// every batch holds the same mix; the queues pull batches from one shared counter (a fast device simply takes more of them).
// Banded global / extension jobs: the read-sets with ragged read ends (msa_device.h msa_device_set_is_ragged) of a batch become a batch of their own -- the
// uniform sets then keep the all-rounds kernel (narrow bands: one launch for all rounds, ~1.4x the lock-step launches' rate on 1 kb reads), which a job with a
// single ragged set would lose for all of them.  (ABPOA_HIP_NO_RAGGED_SPLIT=1: one batch, as before round 5.)
void split_ragged(std::vector<std::vector<int>> &batches, const abpoa_hip_scoring_t *sc, const abpoa_hip_readset_t *sets) {
    if (!sc || sc->wb < 0 || sc->align_mode == ABPOA_HIP_LOCAL_MODE || env_on("ABPOA_HIP_NO_RAGGED_SPLIT")) return;
    std::vector<std::vector<int>> out_;
    for (auto &b_ : batches) {
        std::vector<int> uni, rag;
        for (int i : b_) (msa_device_set_is_ragged(sets[i]) ? rag : uni).push_back(i);
        if (uni.empty() || rag.empty()) { out_.push_back(std::move(b_)); continue; }
        out_.push_back(std::move(uni)); out_.push_back(std::move(rag));
    }
    batches.swap(out_);
}
std::vector<std::vector<int>> deal_batches(const abpoa_hip_scoring_t *sc, const abpoa_hip_readset_t *sets, int n_sets, int n_q) {
    std::vector<std::vector<int>> batches;
    if (n_q == 1) {
        batches.emplace_back(n_sets);
        for (int s = 0; s < n_sets; ++s) batches[0][s] = s;
        split_ragged(batches, sc, sets);
        return batches;
    }
    std::vector<int64_t> cost(n_sets);
    for (int s = 0; s < n_sets; ++s) {
        int64_t sum = 0;
        for (int r = 0; r < sets[s].n_reads; ++r) sum += sets[s].lens[r];
        cost[s] = sum * std::max(1, sets[s].n_reads);
    }
    std::vector<int> order(n_sets);
    for (int s = 0; s < n_sets; ++s) order[s] = s;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return cost[a] > cost[b]; });
    const int per_q = std::max(1, env_int("ABPOA_GPU_BATCHES_PER_DEVICE", 2));
    // a batch should hold >= 1024 sets when the job allows: the device kernels run one wavefront per read-set, a GPU has 1024 SIMDs, and the
    // all-rounds kernel of the narrow-band jobs is at its best with ~1000 resident sets; never fewer batches than queues
    int nb = std::max(n_q, std::min(n_q * per_q, n_sets / 1024));
    nb = std::max(1, std::min(nb, n_sets));
    batches.resize(nb);
    for (int i = 0; i < n_sets; ++i) batches[i % nb].push_back(order[i]);
    for (auto &b_ : batches) std::sort(b_.begin(), b_.end());        // (caller order inside a batch)
    split_ragged(batches, sc, sets);
    return batches;
}
}  // namespace
}  // namespace abpoa_hip

// A context (include/abpoa_hip.h abpoa_hip_ctx_t): the per-caller state of the batch entry -- device, device queue (pool cache, stream, the
// all-rounds kernel's argument record), timing and last error of its own calls -- so that several host threads can run batches side by side.
struct abpoa_hip_ctx { int device; int slot; abpoa_hip_msa_timing_t timing; char err[512]; };
namespace abpoa_hip {
namespace {
std::mutex g_ctx_mu;
bool g_ctx_slot_used[MSA_DEVICE_SLOTS] = {false};
constexpr int CTX_SLOT_LO = MSA_DEVICE_SLOTS / 2;      // the upper half of the device queues belongs to contexts, the lower half to the process-wide entry

// The batch entry proper.  `tm` receives the call's timing record; ctx_slot >= 0: one device queue, the context's, on ctx_device.
int msa_batch_impl(const abpoa_hip_scoring_t *sc_in, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, unsigned flags,
                   int n_threads, abpoa_hip_msa_timing_t &tm, int ctx_device, int ctx_slot) {
    if (engine_device() < 0) { const int rc = abpoa_hip_init(ctx_device >= 0 ? ctx_device : 0); if (rc) return rc; }
    abpoa_hip_scoring_t sc_norm;
    const abpoa_hip_scoring_t *sc = sc_in;
    if (sc_in && sc_in->align_mode == ABPOA_HIP_LOCAL_MODE) { sc_norm = *sc_in; sc_norm.wb = -1; sc = &sc_norm; }      // ref abpoa_align.c:150: local mode has no band
    if (n_sets > 0 && sc && sets && out && msa_device_eligible(sc, flags)) {
        // device-resident driver first; sets that outgrow a device capacity (and whole jobs that do not fit) go to the host driver
        if (n_threads <= 0) n_threads = effective_host_cores();
        for (int s = 0; s < n_sets; ++s) {
            if (sets[s].n_reads < 0) return ABPOA_HIP_EINVAL;
            for (int r = 0; r < sets[s].n_reads; ++r)
                if (sets[s].lens[r] <= 0 || !sets[s].seqs[r]) { set_err("read-set %d: read %d is empty", s, r); return ABPOA_HIP_EINVAL; }
        }
        for (int s = 0; s < n_sets; ++s) memset(&out[s], 0, sizeof(out[s]));
        std::vector<int> devs = device_list();
        if ((int)devs.size() > CTX_SLOT_LO) devs.resize(CTX_SLOT_LO);
        if (ctx_slot >= 0) devs.assign(1, ctx_device);      // a context: its own device, its own queue
        int n_q = (int)devs.size();
        const std::vector<std::vector<int>> batches = deal_batches(sc, sets, n_sets, n_q);
        // (experiment, ABPOA_HIP_RAGGED_CONCURRENT=1: the ragged batch of a mixed job on a second queue of the same device, beside the uniform batch's all-rounds kernel)
        if (n_q == 1 && ctx_slot < 0 && batches.size() == 2 && env_on("ABPOA_HIP_RAGGED_CONCURRENT")) { devs.push_back(devs[0]); n_q = 2; }
        std::atomic<int> next{0};
        std::vector<PassOut> results(batches.size());
        std::vector<double> q_busy(n_q, 0.0);
        auto worker = [&](int q) {
            const int thr = std::max(1, n_threads / n_q);
            for (int b_; (b_ = next.fetch_add(1)) < (int)batches.size();) {
                const auto t0 = std::chrono::steady_clock::now();
                results[b_] = device_passes(sc, sets, out, batches[b_], thr, devs[q], ctx_slot >= 0 ? ctx_slot : q, flags);
                q_busy[q] += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                if (results[b_].rc != ABPOA_HIP_OK) break;
            }
        };
        std::vector<std::thread> th;
        for (int q = 1; q < n_q; ++q) th.emplace_back(worker, q);
        worker(0);
        for (auto &t : th) t.join();
        (void)hipSetDevice(engine_device());
        DeviceRunStats tot;
        memset(&tot, 0, sizeof(tot));
        int rc_dev = ABPOA_HIP_OK;
        std::vector<int> todo;
        for (size_t b_ = 0; b_ < batches.size(); ++b_) {
            const PassOut &R = results[b_];
            if (R.rc != ABPOA_HIP_OK && rc_dev == ABPOA_HIP_OK) rc_dev = R.rc;
            // a batch the device path could not take (it does not fit even alone, or its shape is not the device driver's): only THAT batch's
            // open sets go to the host driver -- R.left holds them -- the finished results of the other batches stay
            todo.insert(todo.end(), R.left.begin(), R.left.end());
            add_stats(tot, R.tot);
        }
        {   // why those sets left the device (the last pass each of them was in; 11: its batch as a whole was not the device's)
            std::lock_guard<std::mutex> lk(g_reason_mu);
            memset(g_host_reasons, 0, sizeof(g_host_reasons));
            for (const PassOut &R : results) for (int i_ : R.left) { auto it = R.why.find(i_); g_host_reasons[it == R.why.end() ? 11 : std::min(std::max(it->second, 0), 10)]++; }
        }
        if (rc_dev != ABPOA_HIP_OK) { free_all(out, n_sets); return rc_dev; }
        if (n_q > 1) tot.device_s = tot.total_s = *std::max_element(q_busy.begin(), q_busy.end());      // queues ran side by side: the busiest one is the wall time
        if (n_q > 1 && opt_env("ABPOA_HIP_VERBOSE")) {
            fprintf(stderr, "[abpoa-hip] %d device queues, %zu batches; busy seconds per queue:", n_q, batches.size());
            for (int q = 0; q < n_q; ++q) fprintf(stderr, " dev%d %.3f", devs[q], q_busy[q]);
            fprintf(stderr, "\n");
        }
        std::sort(todo.begin(), todo.end());
        StreamStats ss;
        ss.n_launches = tot.n_rounds; ss.n_alignments = tot.n_alignments; ss.n_cells = tot.n_cells; ss.algo_bytes = tot.algo_bytes;
        ss.kernel_ms = tot.rows_ms; ss.tail_ms = tot.tail_ms;
        ss.rounds_ms = tot.rounds_ms; ss.rounds_launches = tot.rounds_launches; ss.rounds_algo_bytes = tot.rounds_algo_bytes;
        add_global_stats(ss);
        memset(&tm, 0, sizeof(tm));
        tm.engine_s = tot.device_s; tm.cons_s = tot.cons_s; tm.total_s = tot.total_s;
        tm.n_rounds = tot.n_rounds; tm.n_threads = n_threads; tm.n_groups = n_q;
        tm.host_sort_s = tot.prepare_ms / 1e3; tm.host_fuse_s = tot.fuse_ms / 1e3;      // device kernels now: graph -> rows, cigar -> graph
        tm.n_host_sets = (int32_t)todo.size();                                          // how many sets take the host driver
        if (todo.empty()) return ABPOA_HIP_OK;
        if (opt_env("ABPOA_HIP_VERBOSE"))
            fprintf(stderr, "[abpoa-hip] %zu of %d read-sets outgrew a device capacity: host driver for those\n", todo.size(), n_sets);
        if (strict_mode()) {
            free_all(out, n_sets);
            set_err("ABPOA_HIP_STRICT: %zu of %d read-sets would take the host driver (device capacities: node / edge / aligned slots, arena)",
                    todo.size(), n_sets);
            return ABPOA_HIP_ESTRICT;
        }
        std::vector<abpoa_hip_readset_t> sub(todo.size());
        std::vector<abpoa_hip_msa_t> sub_out(todo.size());
        for (size_t i = 0; i < todo.size(); ++i) sub[i] = sets[todo[i]];
        abpoa_hip_msa_timing_t t2;
        const int rc2 = run_msa_batch(sc, (int)todo.size(), sub.data(), sub_out.data(), flags, n_threads, 0, make_hip_aligner, &t2);
        if (rc2 != ABPOA_HIP_OK) { free_all(out, n_sets); return rc2; }
        for (size_t i = 0; i < todo.size(); ++i) out[todo[i]] = sub_out[i];
        return ABPOA_HIP_OK;
    }
    // the whole job on the host driver: its options are not the device-resident driver's (linear gaps, extension mode, -s, no band in
    // global mode, local reads beyond the local row loop)
    if (strict_mode() && n_sets > 0) {
        set_err("ABPOA_HIP_STRICT: this job's options are the host driver's (see msa_device_eligible)");
        return ABPOA_HIP_ESTRICT;
    }
    const int rc_host = run_msa_batch(sc, n_sets, sets, out, flags, n_threads, 0, make_hip_aligner, &tm);
    tm.n_host_sets = n_sets;
    { std::lock_guard<std::mutex> lk(g_reason_mu); memset(g_host_reasons, 0, sizeof(g_host_reasons)); g_host_reasons[11] = n_sets; }
    return rc_host;
}
}  // namespace
}  // namespace abpoa_hip

extern "C" {
void abpoa_hip_get_host_reasons(int32_t *out12) {
    std::lock_guard<std::mutex> lk(abpoa_hip::g_reason_mu);
    for (int i = 0; i < abpoa_hip::MSA_HOST_REASONS; ++i) out12[i] = abpoa_hip::g_host_reasons[i];
}
int abpoa_hip_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, unsigned flags, int n_threads) {
    abpoa_hip::refresh_options();
    return abpoa_hip::msa_batch_impl(sc, n_sets, sets, out, flags, n_threads, abpoa_hip::g_timing, -1, -1);
}
void abpoa_hip_get_msa_timing(abpoa_hip_msa_timing_t *out) { *out = abpoa_hip::g_timing; }
// ---- contexts
abpoa_hip_ctx_t *abpoa_hip_ctx_create(int device) {
    using namespace abpoa_hip;
    if (engine_device() < 0) { if (abpoa_hip_init(device >= 0 ? device : 0) != ABPOA_HIP_OK) return nullptr; }
    int n = 0; (void)hipGetDeviceCount(&n);
    if (device < 0) device = engine_device();
    if (device >= n) { set_err("device %d out of range (0..%d)", device, n - 1); return nullptr; }
    std::lock_guard<std::mutex> lk(g_ctx_mu);
    for (int s = MSA_DEVICE_SLOTS - 1; s >= CTX_SLOT_LO; --s) if (!g_ctx_slot_used[s]) {
        g_ctx_slot_used[s] = true;
        abpoa_hip_ctx *c = new abpoa_hip_ctx(); c->device = device; c->slot = s; memset(&c->timing, 0, sizeof(c->timing)); c->err[0] = 0;
        return c;
    }
    set_err("all %d batch contexts are in use", MSA_DEVICE_SLOTS - CTX_SLOT_LO);
    return nullptr;
}
void abpoa_hip_ctx_destroy(abpoa_hip_ctx_t *c) {
    if (!c) return;
    { std::lock_guard<std::mutex> lk(abpoa_hip::g_ctx_mu); abpoa_hip::g_ctx_slot_used[c->slot] = false; }
    delete c;      // (the queue's pools stay cached for the next context that takes the slot; abpoa_hip_trim releases them)
}
int abpoa_hip_msa_batch_ctx(abpoa_hip_ctx_t *c, const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, unsigned flags, int n_threads) {
    abpoa_hip::refresh_options();
    if (!c) return ABPOA_HIP_EINVAL;
    abpoa_hip::clear_thread_error();
    const int rc = abpoa_hip::msa_batch_impl(sc, n_sets, sets, out, flags, n_threads, c->timing, c->device, c->slot);
    if (rc != ABPOA_HIP_OK) { const char *m = abpoa_hip::thread_last_error(); snprintf(c->err, sizeof(c->err), "%s", m[0] ? m : abpoa_hip_last_error()); } else c->err[0] = 0;
    return rc;
}
void abpoa_hip_ctx_get_msa_timing(const abpoa_hip_ctx_t *c, abpoa_hip_msa_timing_t *out) { if (c && out) *out = c->timing; }
const char *abpoa_hip_ctx_last_error(const abpoa_hip_ctx_t *c) { return c ? c->err : "null context"; }
void abpoa_hip_trim(void) { abpoa_hip::release_msa_device_caches(); }
}
