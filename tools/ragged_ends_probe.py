"""How the device-resident driver fares on reads that do not start / end at the same place (amplicon-like vs window-cut data): every read is a random
substring of its noisy full-length version; prints how many sets left the device (abpoa_hip_msa_timing_t.n_host_sets) and why.
usage: python tools/ragged_ends_probe.py [max fraction cut from each end, default 0.1]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, ffi, synth

frac = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
rng = np.random.default_rng(5)
for n_reads, length in ((20, 1000), (50, 1000), (50, 3000)):
    sets = []
    for i in range(64):
        reads = list(synth.make_read_set(7, i, n_reads, length, 0.05))
        out = [reads[0]]
        for r in reads[1:]:
            a = int(rng.integers(0, int(frac * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(frac * len(r)) + 1))
            out.append(r[a:b])
        sets.append(out)
    r = api.msa_batch(sets, api.Params(), n_threads=8)
    print(f"{n_reads} reads x {length} bases, up to {frac:.0%} cut per end: {api.msa_timing()['n_host_sets']} of {len(sets)} sets left the device", flush=True)
