"""A job of uniform read-sets with a few ragged ones among them (50 x 1 kb, 5 %; every tenth set has up to 10 % cut from the ends of its reads): read-sets/s of
abpoa_hip_msa_batch with the ragged sets as a batch of their own (default since round 5: the uniform ones keep the all-rounds kernel) and in one batch
(ABPOA_HIP_NO_RAGGED_SPLIT=1).  usage: python tools/mixed_job_probe.py [n_sets]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, ffi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
rng = np.random.default_rng(11)
sets = []
for i in range(n):
    reads = list(synth.make_read_set(9, i, 50, 1000, 0.05))
    if i % 10 == 3:
        reads = [reads[0]] + [r[int(rng.integers(0, len(r) // 10 + 1)):len(r) - int(rng.integers(0, len(r) // 10 + 1))] for r in reads[1:]]
    sets.append(reads)
p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
enc = api.EncodedSets(sets, p.m)
keep = None
for split in (1, 0, 2):      # 2: the ragged batch on a second queue of the device (ABPOA_HIP_RAGGED_CONCURRENT=1)
    os.environ["ABPOA_HIP_NO_RAGGED_SPLIT"] = "0" if split else "1"; os.environ["ABPOA_HIP_RAGGED_CONCURRENT"] = "1" if split == 2 else "0"
    api.msa_batch(None, p, encoded=enc, n_threads=16)
    t = time.time(); r = api.msa_batch(None, p, encoded=enc, n_threads=16); dt = time.time() - t
    print(f"{('one batch        ', 'ragged sets apart', 'apart, two queues')[split]} {n / dt:8.1f} read-sets/s  n_host_sets {api.msa_timing()['n_host_sets']}  all-rounds launches {ffi.stats()['rounds_launches']}", flush=True)
    cons = [x.cons_seq for x in r]
    if keep is None: keep = cons
    else: print("   same consensus:", keep == cons)
