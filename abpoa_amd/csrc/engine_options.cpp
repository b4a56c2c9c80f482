#include "engine_options.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>

namespace abpoa_hip {
namespace {

struct Row { const char *name, *help; };
// P: product switches (documented in INTEGRATION.md); T: test hooks (force a code path that the engine otherwise picks by itself); D: diagnostics
const Row kTable[] = {
    {"ABPOA_GPU_DEVICES", "P  device ordinals abpoa_hip_msa_batch spreads its batches over: all | 0,1,... (default: the device of abpoa_hip_init)"},
    {"ABPOA_GPU_BATCHES_PER_DEVICE", "P  batches dealt per device queue [2]"},
    {"ABPOA_HIP_STRICT", "P  1: a call that would hand any read-set to the host driver fails with ABPOA_HIP_ESTRICT"},
    {"ABPOA_HIP_VERBOSE", "P  per-pass timings, residency and fallback reasons on stderr"},
    {"ABPOA_HIP_HOSTGRAPH", "T  1: the host driver for every job"},
    {"ABPOA_HIP_NO_DEVICE_STRAND", "T  1: -s jobs on the host driver"},
    {"ABPOA_HIP_NO_DEVICE_LOCAL", "T  1: local-mode jobs on the host driver"},
    {"ABPOA_HIP_NO_DEVICE_GENERAL", "T  1: general-kernel jobs on the host driver"},
    {"ABPOA_HIP_DEVICE_GENERAL", "T  1: the general kernel for every device-resident job"},
    {"ABPOA_HIP_LOCKSTEP", "T  1: one launch per phase and round also where the all-rounds kernel would run"},
    {"ABPOA_HIP_NODIR", "T  1: score-record arenas instead of direction words"},
    {"ABPOA_HIP_DIR_WIDE", "T  1 / 0: direction words for wide-band alignments always / never (default: by residency and memory)"},
    {"ABPOA_HIP_NOFAST", "T  1: no fast row loops (general kernel)"},
    {"ABPOA_HIP_NOWIDE", "T  1: no wide row loop"},
    {"ABPOA_HIP_NOXL", "T  1: no long-read form of the wide row loop"},
    {"ABPOA_HIP_TEAM", "T  1 | 2 | 4 wavefronts per wide-band alignment"},
    {"ABPOA_HIP_LOCAL_TEAM", "T  1 / 0: the four-wavefront local row loop always / never"},
    {"ABPOA_HIP_WIDE_LO", "T  smallest band half-width that takes the wide loop [40]"},
    {"ABPOA_HIP_RING_ROWS", "T  depth of the wide loop's score ring (4 | 8 | 16)"},
    {"ABPOA_HIP_EXTRA_ROUTE_MIN", "T  ragged sets with fewer extra columns keep the narrow loop"},
    {"ABPOA_HIP_NO_RAGGED_SPLIT", "T  1: read-sets with ragged read ends stay in the batch of the uniform ones (no all-rounds kernel for such a job)"},
    {"ABPOA_HIP_RAGGED_CONCURRENT", "T  1: the ragged batch of a mixed job runs on a second queue of the device beside the uniform batch (measured: see LOG.md)"},
    {"ABPOA_HIP_PASS_SETS", "T  read-sets per pass of the device-resident driver"},
    {"ABPOA_HIP_FIRST_PASS", "T  start the node-slot ladder at pass 1 / 2 / 3 (profiling runs of one step)"},
    {"ABPOA_HIP_NO_PASS_HINT", "T  1: always start the ladder at 3x"},
    {"ABPOA_HIP_GROUPS", "T  read-set groups of the host driver"},
    {"ABPOA_HIP_ARENA_PCT", "T  arena estimate in percent (overflow-retry tests)"},
    {"ABPOA_HIP_BT_BYTES", "T  LDS window of the tail kernel in bytes"},
    {"ABPOA_HIP_BT_WC", "T  columns per row of the tail's column-slice windows"},
    {"ABPOA_HIP_ORDER_LDS", "T  0: the row-order walk with global tables"},
    {"ABPOA_HIP_ORDER_CAP", "T  node capacity of the row-order walk's LDS tables"},
    {"ABPOA_HIP_DBG", "D  bit mask handed to the kernels (DevBatch.dbg: 64 no fast loops, 128 keep counters, 1024 one wavefront per backtrack, 2048 the compiler's tight loop, ...)"},
    {"ABPOA_HIP_DEVSYNC", "D  1: compare device graph / row order / MSA with the host's after every read"},
    {"ABPOA_HIP_DEVICE_AUDIT", "D  1: every pool of a device queue must sit on that queue's device"},
    {"ABPOA_HIP_CIGAR_DIGEST", "D  1: fold every graph cigar into a per-set digest on the device"},
    {"ABPOA_HIP_DIRTRACE", "D  1: keep direction words and scores (word-level tests)"},
    {"ABPOA_HIP_IMBAL", "D  1: per-round mean / slowest alignment"},
    {"ABPOA_HIP_WIDE_COUNTERS", "D  path census of the wide loop (diagnostic build)"},
    {"ABPOA_HIP_ROW_CENSUS", "D  row census of the narrow loop (diagnostic build)"},
    {"ABPOA_HIP_ORDER_PROF", "D  in-kernel clock of the row-order walk (diagnostic build)"},
};
constexpr int N = (int)(sizeof(kTable) / sizeof(kTable[0]));

struct Snap { bool set[N]; std::string val[N]; };
std::mutex g_mu;
bool g_over_on[N]; std::string g_over[N];      // abpoa_hip_set_option
std::atomic<const Snap *> g_snap{nullptr};

int find(const char *name) { for (int i = 0; i < N; ++i) if (strcmp(kTable[i].name, name) == 0) return i; return -1; }

}  // namespace

void refresh_options() {
    std::lock_guard<std::mutex> lk(g_mu);
    Snap s;
    for (int i = 0; i < N; ++i) {
        const char *e = g_over_on[i] ? g_over[i].c_str() : getenv(kTable[i].name);
        s.set[i] = e != nullptr; s.val[i] = e ? e : "";
    }
    const Snap *cur = g_snap.load(std::memory_order_acquire);
    bool same = cur != nullptr;
    for (int i = 0; same && i < N; ++i) same = cur->set[i] == s.set[i] && cur->val[i] == s.val[i];
    // (a changed snapshot is a new object; the old one is left alone: another thread's call may still read it -- switches change between calls of tests, not in production)
    if (!same) g_snap.store(new Snap(s), std::memory_order_release);
}

const char *opt_env(const char *name) {
    const Snap *cur = g_snap.load(std::memory_order_acquire);
    if (!cur) { refresh_options(); cur = g_snap.load(std::memory_order_acquire); }
    const int i = find(name);
    return (i >= 0 && cur->set[i]) ? cur->val[i].c_str() : nullptr;
}

int set_option(const char *name, const char *value) {
    const int i = name ? find(name) : -1;
    if (i < 0) return -1;
    { std::lock_guard<std::mutex> lk(g_mu); g_over_on[i] = value != nullptr; g_over[i] = value ? value : ""; }
    refresh_options();
    return 0;
}

int list_options(const char **names, const char **help, int cap) {
    for (int i = 0; i < N && i < cap; ++i) { if (names) names[i] = kTable[i].name; if (help) help[i] = kTable[i].help; }
    return N;
}

}  // namespace abpoa_hip
