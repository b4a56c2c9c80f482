// Device-resident read-set driver (msa_device.cpp / poa_device.hip): the graph of every read-set stays in HBM for the whole
// progressive alignment; the host only uploads the reads and downloads the finished graphs.
#pragma once
#include <stdint.h>
#include <vector>
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {

struct DeviceRunStats {
    double prepare_ms, rows_ms, tail_ms, fuse_ms;     // summed kernel durations (hipEvents on the job's stream)
    double device_s, cons_s, total_s;                 // wall: first launch -> graphs on the host; consensus; whole call
    int64_t n_cells, algo_bytes, n_alignments; int32_t n_rounds, n_fit_3x;      // n_fit_3x: finished sets whose graph would also have fitted 3x the longest read in node slots
    // all-rounds kernel (poa_rounds.hip), when the job took it: launches, their duration, the share of the sets' clock ticks spent in the row loop, and
    // mean set / slowest set (how much of the kernel's duration the average workgroup was busy); rows_ms .. fuse_ms above then hold the duration split by phase
    double rounds_ms, rounds_rows_share, rounds_mean_over_max; int32_t rounds_launches, pad2;
    int64_t rounds_algo_bytes;     // algorithmic bytes of the cells computed inside the all-rounds kernel
    double rounds_mticks[4];      // mean per set: 10^6 shader-clock ticks in prepare / row loop / backtrack / fuse
};

// true when the scoring / output options can run on the device-resident path: every gap model and alignment mode (the fast row loops for banded global and
// short local alignments, the general kernel for linear gaps, extension mode, global mode without a band and long local reads), consensus and / or MSA
// output, alphabets of up to 27 codes, per-base weights, the strand retry (-s).  env ABPOA_HIP_HOSTGRAPH=1 forces the host driver;
// ABPOA_HIP_NO_DEVICE_LOCAL / _GENERAL / _STRAND=1 send that class of jobs there
bool msa_device_eligible(const abpoa_hip_scoring_t *sc, unsigned flags);
// A read-set whose reads differ much in length (more than an eighth of the longest, at least 64 bases): its band sits that far from the alignment's path in every
// row (reference abpoa_align.h:34-35), so its rows are wider than 2 w by the difference -- run_msa_device gives it `band_extra` columns and the wide row loop.
// A job that has such sets cannot take the all-rounds kernel: abpoa_hip_msa_batch runs them as a batch of their own (msa_hip.cpp deal_batches).
inline bool msa_device_set_is_ragged(const abpoa_hip_readset_t &S) {
    if (S.n_reads < 2) return false;
    int mx = 0, mn = 0x7fffffff;
    for (int r = 0; r < S.n_reads; ++r) { mx = S.lens[r] > mx ? S.lens[r] : mx; mn = S.lens[r] < mn ? S.lens[r] : mn; }
    const int tol = mx / 8 > 64 ? mx / 8 : 64;
    return mx - mn > tol;
}

// Consensus of every set in out[]; sets whose graph outgrew a device capacity are listed in `fallback` (out[] zeroed for
// them) and must be redone (with a larger node_factor, or by the host driver; an entry -(s + 1) is set s with a node out of edge slots: more node slots
// would not help, the last pass does).  node_factor: node slots per set = factor x
// longest read; 4096 and more: the last pass -- every node also has an edge slot per read (score records instead of direction words).  ABPOA_HIP_ENOMEM: the job does not fit the device (split it); EINVAL: not a job for the device driver.
// device < 0: the device the engine was initialised on.  slot: which of the per-worker pool caches to use (one worker = one device queue of
// the multi-GPU batch call; workers may share a device); a slot runs one job at a time.
constexpr int MSA_DEVICE_SLOTS = 16;
// Wide-band jobs (10 kb reads: one wavefront per alignment, tens of KB of LDS each): how many read-sets the device holds at once -- workgroups per CU by
// the wide row loop's LDS x CUs.  A launch with more alignments than that runs its workgroups in turns, and because the alignments of a round take
// about the same time the last, partly filled turn costs as much as a full one: the caller cuts such a job into passes of this size.  0: no preference.
int msa_device_resident_sets(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets);
// frees every cached pool of every device queue (abpoa_hip_trim)
void release_msa_device_caches();
int run_msa_device(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out, int n_threads,
                   std::vector<int> *fallback, DeviceRunStats *stats, double node_factor, unsigned flags, int device = -1, int slot = 0,
                   std::vector<int> *fallback_reason = nullptr);
// fallback_reason (optional, parallel to fallback): why the set left the pass -- 1 node slots at the first read, 2 predecessor list slots, 3 cigar slots,
// 4 node slots in the fuse phase, 5 edge / aligned slots of a node, 6 projected node growth, 7 row-order walk, 8 MSA rank walk, 9 DP arena too small for the
// bands, 10 other DP status, 0 other (abpoa_hip_get_host_reasons)
constexpr int MSA_HOST_REASONS = 12;

}  // namespace abpoa_hip
