// Lock-step read-set driver, parameterised on the batch aligner so the host logic can be exercised on CPU by the
// tests (tests/cpu_shim.cpp supplies an oracle-backed GroupAligner; the product supplies the HIP one and nothing else).
#pragma once
#include "../../include/abpoa_hip.h"
#include "batch_types.h"

namespace abpoa_hip {
int run_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out,
                  unsigned flags, int n_threads, int n_groups, AlignerFactory make, abpoa_hip_msa_timing_t *timing);
}
