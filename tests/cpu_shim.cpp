// TEST INFRASTRUCTURE.  Builds the product's HOST sources (poa_graph.cpp, msa_batch.cpp) into a
// CPU-only library whose abpoa_hip_align_batch is backed by the oracle, so that graph fusion, row
// ordering, consensus and MSA can be checked against the reference's outputs without a GPU.
// Never shipped: the product library gets abpoa_hip_align_batch from engine.cpp (HIP) only.
#include <string.h>
#include "../include/abpoa_hip.h"
extern "C" {
#include "../oracle/abpoa_dp_oracle.h"
int abpoa_hip_align_batch(const abpoa_hip_scoring_t *sc, int n, const abpoa_hip_problem_t *pb, abpoa_hip_result_t *res, unsigned flags) {
    (void)flags;
    for (int i = 0; i < n; ++i) { int rc = abpoa_oracle_align(sc, &pb[i], &res[i], 0); if (rc) return rc; }
    return 0;
}
const char *abpoa_hip_last_error(void) { return "cpu shim"; }
}
