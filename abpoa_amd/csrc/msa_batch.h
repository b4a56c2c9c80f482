// Lock-step read-set driver, parameterised on the batch aligner so the host logic can be exercised
// on CPU by the tests (tests/cpu_shim.cpp passes an oracle-backed aligner; the product passes
// abpoa_hip_align_batch and nothing else).
#pragma once
#include "../../include/abpoa_hip.h"

namespace abpoa_hip {
typedef int (*AlignBatchFn)(const abpoa_hip_scoring_t *, int, const abpoa_hip_problem_t *, abpoa_hip_result_t *, unsigned);
int run_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out,
                  unsigned flags, int n_threads, AlignBatchFn align, abpoa_hip_msa_timing_t *timing);
}
