// Read-set batch driver (include/abpoa_hip.h section 3).  Reference counterpart per set:
// abpoa_msa -> abpoa_poa -> {abpoa_align_sequence_to_graph, abpoa_add_graph_alignment} -> abpoa_output
// (src/abpoa_align.c:302-437).  Here the per-read loop is turned inside out: the sets are split into a few groups;
// inside a group all sets advance one read per round so that one engine launch carries one alignment of every set
// of the group, and the groups run out of phase on their own streams so that host graph work (fusion, row ordering,
// flattening straight into pinned staging memory) of one group overlaps the DP kernel of another.
#include <algorithm>
#include <atomic>
#include <stdio.h>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <memory>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <vector>
#include "engine_options.h"
#include "msa_batch.h"
#include "poa_graph.h"

#include <unordered_map>
namespace abpoa_hip {

namespace {
std::mutex g_dig_mu; std::unordered_map<uint64_t, uint64_t> g_dig;
// same arithmetic as poa_device.h (poa_mix64 / poa_cigar_word_mix / poa_cigar_digest_round), restated: this file also builds without HIP (tests/cpu_shim.cpp)
uint64_t digest_mix64(uint64_t z) { z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
uint64_t digest_word_mix(uint64_t word, int i) { return digest_mix64(word + 0xD6E8FEB86659FD93ull * (uint64_t)(i + 1)); }
uint64_t digest_round(uint64_t before, int read_index, int n_cigar, uint64_t sum) { return (before * 0x9E3779B97F4A7C15ull) ^ (sum + digest_mix64(((uint64_t)(uint32_t)read_index
        << 32) | (uint32_t)n_cigar)); }
uint64_t fnv64(const void *p, size_t n, uint64_t h = 1469598103934665603ull) { const uint8_t *b = (const uint8_t *)p; for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ull; } return h; }
}
bool cigar_digest_on() { static const bool on = opt_env("ABPOA_HIP_CIGAR_DIGEST") && atoi(opt_env("ABPOA_HIP_CIGAR_DIGEST")); return on; }
// (the digest function is poa_device.h's: the device driver folds the same per-word mixes in its fuse phase)
void cigar_digest_add(const uint8_t *seq0, int len0, int read_index, const uint64_t *cigar, int n_cigar) {
    const uint64_t key = fnv64(seq0, (size_t)len0);
    uint64_t sum = 0; for (int i = 0; i < n_cigar; ++i) sum += digest_word_mix(cigar[i], i);
    std::lock_guard<std::mutex> lk(g_dig_mu);
    uint64_t &d = g_dig[key]; d = digest_round(d, read_index, n_cigar, sum);      // (order-dependent within a set: reads are folded in order)
}
void cigar_digest_set(const uint8_t *seq0, int len0, uint64_t value) {      // (device driver: the digest its fuse phase kept for the set)
    std::lock_guard<std::mutex> lk(g_dig_mu);
    g_dig[fnv64(seq0, (size_t)len0)] = value;
}

int effective_host_cores() {
    int n = (int)std::thread::hardware_concurrency(); if (n < 1) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64]; long period = 0;
        if (fscanf(f, "%63s %ld", q, &period) == 2 && strcmp(q, "max") != 0 && period > 0) { const long quota = atol(q); if (quota > 0) n = std::min<long>(n, (quota + period - 1) / period); }
        fclose(f);
    }
    return n < 1 ? 1 : n;
}

namespace {
// Minimal persistent fork-join pool: run(n, fn) executes fn(i) for i in [0,n) on all workers + caller.
class Pool {
  public:
    explicit Pool(int n) : n_(n < 1 ? 1 : n) {
        for (int t = 1; t < n_; ++t) th_.emplace_back([this] { worker(); });
    }
    ~Pool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; ++gen_; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    void run(int n, const std::function<void(int)> &fn) {
        if (n <= 0) return;
        if (n_ == 1) { for (int i = 0; i < n; ++i) fn(i); return; }
        fn_ = &fn; total_ = n; next_.store(0); pending_.store(n_ - 1);
        { std::lock_guard<std::mutex> lk(mu_); ++gen_; }
        cv_.notify_all();
        drain();
        std::unique_lock<std::mutex> lk(mu_);
        done_cv_.wait(lk, [this] { return pending_.load() == 0; });
    }
  private:
    void drain() { for (int i; (i = next_.fetch_add(1)) < total_;) (*fn_)(i); }
    void worker() {
        uint64_t seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu_); cv_.wait(lk, [&] { return gen_ != seen; }); seen = gen_; if (stop_) return; }
            drain();
            if (pending_.fetch_sub(1) == 1) { std::lock_guard<std::mutex> lk(mu_); done_cv_.notify_all(); }
        }
    }
    int n_; std::vector<std::thread> th_;
    std::mutex mu_; std::condition_variable cv_, done_cv_;
    uint64_t gen_ = 0; bool stop_ = false;
    const std::function<void(int)> *fn_ = nullptr; int total_ = 0;
    std::atomic<int> next_{0}, pending_{0};
};
double now_s() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

struct GroupTimes { double sort_s = 0, engine_s = 0, fuse_s = 0; int rounds = 0; };
inline const int32_t *weights_of(const abpoa_hip_readset_t &rs, int r) { return rs.weights ? rs.weights[r] : nullptr; }
}  // namespace

// One group of read-sets [s0, s1): the whole progressive POA, lock-step over reads.
static int run_group(const abpoa_hip_scoring_t &scoring, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, int s0, int s1,
                     std::vector<PoaGraph> &graphs, int n_threads, GroupAligner *al, bool with_remain, bool amb_strand, GroupTimes *gt) {
    Pool pool(n_threads);
    int max_reads = 0;
    for (int s = s0; s < s1; ++s) if (sets[s].n_reads > max_reads) max_reads = sets[s].n_reads;
    std::vector<int> active; active.reserve(s1 - s0);
    std::vector<BatchShape> shapes;
    std::atomic<int> fail{0};
    for (int k = 0; k < max_reads; ++k) {
        active.clear();
        for (int s = s0; s < s1; ++s) if (sets[s].n_reads > k && out[s].status == 0) active.push_back(s);
        if (active.empty()) break;
        if (k == 0) {       // first read of every set seeds its graph (reference: abpoa_align_sequence_to_graph returns -1, :186)
            pool.run((int)active.size(), [&](int a) {
                const int s = active[a];
                try { graphs[s].add_alignment(sets[s].seqs[0], sets[s].lens[0], nullptr, 0, 0, weights_of(sets[s], 0)); } catch (...) { fail.store(1); }
            });
            if (fail.load()) return ABPOA_HIP_EINVAL;
            continue;
        }
        const double t0 = now_s();
        shapes.resize(active.size());
        for (size_t a = 0; a < active.size(); ++a) {
            const int s = active[a];
            shapes[a] = BatchShape{graphs[s].n_nodes(), sets[s].lens[k], graphs[s].n_edges(), graphs[s].n_edges()};
        }
        // (-s: the reverse-complement retry starts from the band state the forward DP leaves behind, see below)
        const bool carry_band = amb_strand && scoring.wb >= 0;
        int rc = al->prepare(&scoring, (int)active.size(), shapes.data(), carry_band ? GA_BAND_KEEP : GA_BAND_FRESH);
        if (rc) return rc;
        pool.run((int)active.size(), [&](int a) {
            const int s = active[a];
            try {
                graphs[s].topological_sort(with_remain);
                ProblemSlots sl = al->slots(a);
                memcpy(sl.query, sets[s].seqs[k], sets[s].lens[k]);
                graphs[s].flatten_into(with_remain, sl.row_base, sl.row_node_id, sl.row_remain, sl.pred_off, sl.pred_row, sl.out_off, sl.out_row);
            } catch (...) { fail.store(1); }
        });
        if (fail.load()) return ABPOA_HIP_EINVAL;
        const double t1 = now_s();
        rc = al->run();
        if (rc) return rc;
        double t2 = now_s();
        // ---- ambiguous strand (reference abpoa_poa, src/abpoa_align.c:315-336): reads that score below a third of the best possible are
        //      aligned again as their reverse complement, on the SAME rows (the reference calls the DP without re-sorting); the strand
        //      with the strictly better score goes into the graph.  The forward cigars are saved first: the aligner is run again.
        //      max_pos_left/right are reset only by the topological sort (abpoa_graph.c:303-308), which the retry does not repeat (:329 calls
        //      simd_abpoa_align_sequence_to_graph directly): the retry's adaptive band starts from the bounds the forward pass pushed and can only
        //      widen them (row 0 re-assigns its successors to 1, simd_abpoa_align.c:556-561).
        std::vector<int> retry;
        std::vector<std::vector<uint64_t>> fwd_cig;
        std::vector<std::vector<uint8_t>> rc_seq; std::vector<std::vector<int32_t>> rc_w;
        std::vector<char> use_rc(active.size(), 0);
        if (amb_strand) {
            for (size_t a = 0; a < active.size(); ++a) {
                const int s = active[a];
                if (al->status((int)a) != 0) continue;
                const int qlen = sets[s].lens[k], lim = std::min(qlen, graphs[s].n_nodes() - 2);
                if ((double)al->best_score((int)a) < (double)lim * scoring.max_mat * .3333) retry.push_back((int)a);
            }
        }
        if (!retry.empty()) {
            fwd_cig.resize(active.size()); rc_seq.resize(active.size()); rc_w.resize(active.size());
            std::vector<int> fwd_score(active.size(), 0), fwd_st(active.size(), 0); std::vector<int64_t> fwd_cells(active.size(), 0);
            for (size_t a = 0; a < active.size(); ++a) {
                fwd_st[a] = al->status((int)a); fwd_score[a] = al->best_score((int)a); fwd_cells[a] = al->n_cells((int)a);
                fwd_cig[a].assign(al->cigar((int)a), al->cigar((int)a) + al->n_cigar((int)a));
            }
            std::vector<BatchShape> sh2(retry.size());
            std::vector<std::vector<int32_t>> fwd_l(retry.size()), fwd_r(retry.size());
            for (size_t t = 0; t < retry.size(); ++t) {
                sh2[t] = shapes[retry[t]];
                if (carry_band) { const int gn = sh2[t].n_rows; fwd_l[t].assign(al->left(retry[t]), al->left(retry[t]) + gn); fwd_r[t].assign(al->right(retry[t]), al->right(retry[t]) + gn); }
            }
            rc = al->prepare(&scoring, (int)retry.size(), sh2.data(), carry_band ? GA_BAND_SEEDED : GA_BAND_FRESH);
            if (rc) return rc;
            pool.run((int)retry.size(), [&](int t) {
                const int a = retry[t], s = active[a], qlen = sets[s].lens[k];
                try {
                    const uint8_t *q = sets[s].seqs[k]; const int32_t *wq = weights_of(sets[s], k);
                    rc_seq[a].resize(qlen); if (wq) rc_w[a].resize(qlen);
                    for (int j = 0; j < qlen; ++j) { const uint8_t c = q[qlen - j - 1]; rc_seq[a][j] = c < 4 ? (uint8_t)(3 - c) : (uint8_t)4; if (wq) rc_w[a][j] = wq[qlen - j - 1]; }
                    ProblemSlots sl = al->slots(t);
                    memcpy(sl.query, rc_seq[a].data(), qlen);
                    graphs[s].flatten_into(with_remain, sl.row_base, sl.row_node_id, sl.row_remain, sl.pred_off, sl.pred_row, sl.out_off, sl.out_row);
                    if (carry_band) { memcpy(sl.left, fwd_l[t].data(), 4 * fwd_l[t].size()); memcpy(sl.right, fwd_r[t].data(), 4 * fwd_r[t].size()); }
                } catch (...) { fail.store(1); }
            });
            if (fail.load()) return ABPOA_HIP_EINVAL;
            rc = al->run();
            if (rc) return rc;
            t2 = now_s();
            pool.run((int)active.size(), [&](int a) {
                const int s = active[a];
                if (fwd_st[a] != 0) { out[s].status = fwd_st[a]; return; }
                out[s].n_cells += fwd_cells[a];
                const uint64_t *cg = fwd_cig[a].data(); int ncg = (int)fwd_cig[a].size();
                const uint8_t *q = sets[s].seqs[k]; const int32_t *wq = weights_of(sets[s], k);
                const auto it = std::find(retry.begin(), retry.end(), a);
                if (it != retry.end()) {
                    const int t = (int)(it - retry.begin());
                    if (al->status(t) != 0) { out[s].status = al->status(t); return; }
                    out[s].n_cells += al->n_cells(t);
                    if (al->best_score(t) > fwd_score[a]) { cg = al->cigar(t); ncg = al->n_cigar(t); q = rc_seq[a].data(); wq = wq ? rc_w[a].data() : nullptr; if (out[s].is_rc) out[s].is_rc[k] = 1; }
                }
                try { graphs[s].add_alignment(q, sets[s].lens[k], cg, ncg, k, wq); } catch (...) { fail.store(1); }
            });
        } else
        pool.run((int)active.size(), [&](int a) {
            const int s = active[a];
            const int st = al->status(a);
            if (st != 0) { out[s].status = st; return; }
            out[s].n_cells += al->n_cells(a);
            if (cigar_digest_on()) cigar_digest_add(sets[s].seqs[0], sets[s].lens[0], k, al->cigar(a), al->n_cigar(a));
            try { graphs[s].add_alignment(sets[s].seqs[k], sets[s].lens[k], al->cigar(a), al->n_cigar(a), k, weights_of(sets[s], k)); } catch (...) { fail.store(1); }
        });
        if (fail.load()) return ABPOA_HIP_EINVAL;
        const double t3 = now_s();
        gt->sort_s += t1 - t0; gt->engine_s += t2 - t1; gt->fuse_s += t3 - t2; gt->rounds += 1;
    }
    return ABPOA_HIP_OK;
}

int run_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out,
                  unsigned flags, int n_threads, int n_groups, AlignerFactory make, abpoa_hip_msa_timing_t *tm) {
    if (n_sets < 0 || !sc || (n_sets > 0 && (!sets || !out))) return ABPOA_HIP_EINVAL;
    const double t_start = now_s();
    if (tm) memset(tm, 0, sizeof(*tm));
    for (int s = 0; s < n_sets; ++s) memset(&out[s], 0, sizeof(out[s]));
    if (n_sets == 0) return ABPOA_HIP_OK;
    if (n_threads <= 0) n_threads = effective_host_cores();
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_sets) n_threads = n_sets;
    if (n_groups <= 0) { const char *e_ = opt_env("ABPOA_HIP_GROUPS"); if (e_) n_groups = atoi(e_); }
    if (n_groups <= 0) n_groups = n_sets >= 512 ? 4 : (n_sets >= 128 ? 2 : 1);
    if (n_groups > n_threads) n_groups = n_threads;
    const bool want_msa = flags & ABPOA_HIP_OUT_MSA, want_cons = (flags & ABPOA_HIP_OUT_CONS) || !want_msa;
    const bool amb_strand = flags & ABPOA_HIP_AMB_STRAND;      // (any alphabet, as the reference: codes 0..3 are complemented, every other code becomes 4, src/abpoa_align.c:318-321)
    abpoa_hip_scoring_t scoring = *sc;
    if (sc->align_mode == ABPOA_HIP_LOCAL_MODE) scoring.wb = -1;        // reference abpoa_post_set_para, abpoa_align.c:150
    scoring.ret_cigar = 1; scoring.rev_cigar = 0;
    const bool with_remain = scoring.wb >= 0 || scoring.zdrop > 0;
    for (int s = 0; s < n_sets; ++s) {
        if (sets[s].n_reads < 0) return ABPOA_HIP_EINVAL;
        for (int r = 0; r < sets[s].n_reads; ++r) if (sets[s].lens[r] <= 0 || !sets[s].seqs[r]) return ABPOA_HIP_EINVAL;
        out[s].n_reads = sets[s].n_reads;
        if (amb_strand) out[s].is_rc = (uint8_t *)calloc((size_t)std::max(1, sets[s].n_reads), 1);
    }
    std::vector<PoaGraph> graphs(n_sets);
    for (int s = 0; s < n_sets; ++s) graphs[s].reset(sets[s].n_reads, want_msa);

    // ---- groups run concurrently, each with its own aligner (stream) and worker pool
    std::vector<std::unique_ptr<GroupAligner>> aligners;
    for (int g = 0; g < n_groups; ++g) { aligners.emplace_back(make()); if (!aligners.back()) return ABPOA_HIP_ENODEV; }
    std::vector<GroupTimes> gts(n_groups); std::vector<int> rcs(n_groups, 0);
    std::vector<std::thread> drivers;
    const int per_group_threads = n_threads / n_groups > 0 ? n_threads / n_groups : 1;
    for (int g = 0; g < n_groups; ++g) {
        const int s0 = (int)((int64_t)n_sets * g / n_groups), s1 = (int)((int64_t)n_sets * (g + 1) / n_groups);
        auto body = [&, g, s0, s1] { rcs[g] = run_group(scoring, sets, out, s0, s1, graphs, per_group_threads, aligners[g].get(), with_remain, amb_strand, &gts[g]); };
        if (g + 1 < n_groups) drivers.emplace_back(body); else body();
    }
    for (auto &t : drivers) t.join();
    int rc = ABPOA_HIP_OK;
    for (int g = 0; g < n_groups; ++g) if (rcs[g] != 0 && rc == 0) rc = rcs[g];
    aligners.clear();
    if (tm) for (int g = 0; g < n_groups; ++g) {
        tm->host_sort_s += gts[g].sort_s / n_groups; tm->engine_s += gts[g].engine_s / n_groups; tm->host_fuse_s += gts[g].fuse_s / n_groups;
        if (gts[g].rounds > tm->n_rounds) tm->n_rounds = gts[g].rounds;
    }
    if (rc == ABPOA_HIP_OK) {
        const double t0 = now_s();
        Pool pool(n_threads);
        pool.run(n_sets, [&](int s) {
            if (out[s].status != 0 || graphs[s].empty()) return;
            abpoa_hip_msa_t &o = out[s];
            std::vector<int> ids, cov; std::vector<uint8_t> bases;
            if (want_cons) {
                graphs[s].consensus(&ids, &bases, &cov);
                o.cons_len = (int)ids.size();
                o.cons_base = (uint8_t *)malloc(ids.size() + 1); o.cons_cov = (int32_t *)malloc(4 * (ids.size() + 1)); o.cons_node_id = (int32_t *)malloc(4 * (ids.size() + 1));
                memcpy(o.cons_base, bases.data(), bases.size()); memcpy(o.cons_cov, cov.data(), 4 * cov.size()); memcpy(o.cons_node_id, ids.data(), 4 * ids.size());
            }
            if (want_msa) {
                int msa_len = 0; std::vector<std::vector<uint8_t>> rows; std::vector<int> col;
                graphs[s].rc_msa(scoring.m, &msa_len, &rows, &col);
                o.msa_len = msa_len; o.msa_rows = sets[s].n_reads + (want_cons ? 1 : 0);
                o.msa_base = (uint8_t *)malloc((size_t)o.msa_rows * (msa_len > 0 ? msa_len : 1));
                for (int r = 0; r < sets[s].n_reads; ++r) memcpy(o.msa_base + (size_t)r * msa_len, rows[r].data(), msa_len);
                if (want_cons) {                                      // reference abpoa_output.c:151-164
                    uint8_t *crow = o.msa_base + (size_t)sets[s].n_reads * msa_len;
                    memset(crow, scoring.m, msa_len);
                    for (size_t i = 0; i < ids.size(); ++i) crow[col[ids[i]]] = bases[i];
                }
            }
        });
        if (tm) tm->cons_s = now_s() - t0;
    }
    if (tm) { tm->total_s = now_s() - t_start; tm->n_threads = n_threads; tm->n_groups = n_groups; }
    if (rc != ABPOA_HIP_OK) for (int s = 0; s < n_sets; ++s) abpoa_hip_free_msa(&out[s]);
    return rc;
}

}  // namespace abpoa_hip

extern "C" {
// digest of the cigars folded so far for the read-set whose first read is seq0 (0: none); reset = 1 clears the table afterwards
unsigned long long abpoa_hip__cigar_digest(const uint8_t *seq0, int len0, int reset) {
    using namespace abpoa_hip;
    std::lock_guard<std::mutex> lk(g_dig_mu);
    const auto it = g_dig.find(fnv64(seq0, (size_t)len0));
    const unsigned long long v = it == g_dig.end() ? 0ull : it->second;
    if (reset) g_dig.clear();
    return v;
}
void abpoa_hip_free_msa_array(abpoa_hip_msa_t *r, int n) { for (int i = 0; r && i < n; ++i) abpoa_hip_free_msa(&r[i]); }
void abpoa_hip_free_msa(abpoa_hip_msa_t *r) {
    if (!r) return;
    free(r->cons_base); free(r->cons_cov); free(r->cons_node_id); free(r->msa_base); free(r->is_rc);
    r->cons_base = nullptr; r->cons_cov = nullptr; r->cons_node_id = nullptr; r->msa_base = nullptr; r->is_rc = nullptr;
    r->cons_len = r->msa_len = r->msa_rows = 0;
}
}
