// Diagnostics of the device-resident driver (ABPOA_HIP_DEVSYNC=1): synchronise after every kernel and, for the first read-sets of the job, compare what the
// device holds with the host layer fed the same cigars -- row order, graph, MSA, consensus -- plus the per-round load-balance reports (ABPOA_HIP_IMBAL).
// Not on any production path: run_msa_device (msa_device.cpp) constructs one and every method returns at once unless the switch is set.
#pragma once
#include <vector>
#include <hip/hip_runtime.h>
#include "../../include/abpoa_hip.h"
#include "poa_device.h"
#include "poa_graph.h"

namespace abpoa_hip {

// Heaviest-bundling consensus from flat [node][out_cap] edge arrays (the host-side check of poa_consensus_kernel)
void consensus_flat(int n, const int32_t *order, const uint8_t *base, const uint8_t *nout, const int32_t *out_id, const int32_t *out_w, const int32_t *n_read,
                    std::vector<int> *ids, std::vector<uint8_t> *bases, std::vector<int> *cov, std::vector<int> &score, std::vector<int> &max_out, int out_cap);

class DeviceDebug {
public:
    // p: the driver's kernel-argument record (its pointers stay valid for the run); ps: the per-set table
    DeviceDebug(const PoaDev *p, const std::vector<PoaSet> *ps, const abpoa_hip_readset_t *sets, int n_sets, int m, int aln_cap, int max_reads, bool want_msa, bool amb,
                hipStream_t stream);
    bool on() const { return on_; }
    int stage(const char *what, int k);                     // synchronise and report; != 0: the stream is in error
    void order_check(int k);                                // row order of round k against the host graph's Kahn walk (order_mode 1)
    void graph_check(int k);                                // after the fuse of round k: device graph against the host graph fed the same cigar
    void balance_report(int k, const DevBatch &b, hipEvent_t rows_begin, hipEvent_t rows_end);      // ABPOA_HIP_IMBAL: slowest / mean alignment of the round, censuses
    void msa_check(const PoaState *hs, const abpoa_hip_msa_t *out);
    void consensus_check(const PoaState *hs, const abpoa_hip_msa_t *out);
private:
    bool on_; const PoaDev *p_; const std::vector<PoaSet> *ps_; const abpoa_hip_readset_t *sets_; int n_sets_, m_, aln_cap_, max_reads_; bool want_msa_, amb_; hipStream_t st_;
    std::vector<PoaGraph> graphs_;      // host graphs of the first (up to four) sets
    std::vector<double> tot_set_; double sum_max_ = 0;      // balance_report: what lock-step costs over the rounds
};

}  // namespace abpoa_hip
