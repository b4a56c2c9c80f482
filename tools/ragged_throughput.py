"""Read-sets/s of the device-resident driver vs the host driver on reads with ragged ends (tools/ragged_ends_probe.py's inputs): 256 sets x 50 reads x 1 kb,
up to 10 % cut from each end.  usage: python tools/ragged_throughput.py"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, ffi, synth

lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
rng = np.random.default_rng(5)
sets = []
for i in range(256):
    reads = list(synth.make_read_set(7, i, 50, 1000, 0.05))
    out = [reads[0]]
    for r in reads[1:]:
        a = int(rng.integers(0, int(0.1 * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(0.1 * len(r)) + 1))
        out.append(r[a:b])
    sets.append(out)
full = [list(synth.make_read_set(7, i, 50, 1000, 0.05)) for i in range(256)]
p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
for name, data in (("ragged", sets), ("full-length", full)):
    enc = api.EncodedSets(data, p.m)
    keep = None
    for host in (0, 1):
        os.environ["ABPOA_HIP_HOSTGRAPH"] = str(host)
        api.msa_batch(None, p, encoded=enc, n_threads=16)
        t = time.time(); r = api.msa_batch(None, p, encoded=enc, n_threads=16); dt = time.time() - t
        print(f"{name:12s} {'host driver' if host else 'device     '} {len(data) / dt:8.1f} read-sets/s  n_host_sets {api.msa_timing()['n_host_sets']}", flush=True)
        cons = [x.cons_seq for x in r]
        if keep is None: keep = cons
        else: print("   same consensus:", keep == cons)
