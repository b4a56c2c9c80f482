// Lock-step read-set driver, parameterised on the batch aligner so the host logic can be exercised on CPU by the
// tests (tests/cpu_shim.cpp supplies an oracle-backed GroupAligner; the product supplies the HIP one and nothing else).
#pragma once
#include "../../include/abpoa_hip.h"
#include "batch_types.h"

namespace abpoa_hip {
// Test hook (ABPOA_HIP_CIGAR_DIGEST=1): both read-set drivers fold every alignment's graph cigar into a digest per read-set, keyed by a hash of the set's
// first read, so that a test can compare the device-resident driver's cigars with the oracle-backed host run directly (abpoa_hip__cigar_digest).
bool cigar_digest_on();
void cigar_digest_add(const uint8_t *seq0, int len0, int read_index, const uint64_t *cigar, int n_cigar);
void cigar_digest_set(const uint8_t *seq0, int len0, uint64_t value);      // (the device-resident driver keeps the digest on the device: poa_device.h)
int run_msa_batch(const abpoa_hip_scoring_t *sc, int n_sets, const abpoa_hip_readset_t *sets, abpoa_hip_msa_t *out,
                  unsigned flags, int n_threads, int n_groups, AlignerFactory make, abpoa_hip_msa_timing_t *timing);
}
