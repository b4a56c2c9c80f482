#!/usr/bin/env python3
"""Does a second wavefront per SIMD pay for the wide row loop?  5 kb reads (5 % error, affine, 20 per set): with a 4-row score ring and direction words
(ABPOA_HIP_RING_ROWS=4 ABPOA_HIP_DIR_WIDE=1) an alignment needs 19.7 KB of LDS, so eight workgroups share a CU -- two wavefronts per SIMD with 2048
read-sets in flight, one with 1024.  Prints read-sets/s for both (same kernels, same per-alignment work).
usage (GPU box): ABPOA_HIP_RING_ROWS=4 ABPOA_HIP_DIR_WIDE=1 ABPOA_HIP_VERBOSE=1 python3 tools/two_waves_probe.py [length] [reads]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, synth
from abpoa_amd.workloads import WORKLOADS

length = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 20
import multiprocessing as mp
def gen(i):
    return synth.make_read_set(7, i, n_reads=n_reads, length=length, err=0.05, alphabet=synth.NT)
with mp.get_context("fork").Pool(16) as pool:
    sets = pool.map(gen, range(2048))
par = api.Params(**WORKLOADS["cfg4"]["params"])
api.msa_batch(sets[:64], par)
for n in (1024, 2048, 1024, 2048):
    t0 = time.time(); r = api.msa_batch(sets[:n], par); dt = time.time() - t0
    assert all(x.status == 0 for x in r)
    print(f"{n} read-sets x {n_reads} x {length} b: {dt:.3f} s, {n / dt:.1f} read-sets/s", flush=True)
