"""bench.py's `cfg2_ragged_x1024` job alone (same seed, same sets), for a profiler: two timed calls of the batch entry, rate and the driver's phase clocks.
usage: [rocprofv3 --kernel-trace --stats -d DIR --] python3 tools/ragged_profile.py [n_sets]"""
import json
import os
import sys
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, synth
from abpoa_amd.workloads import WORKLOADS

n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
rng = np.random.default_rng(20240)
sets = []
for i in range(n_sets):
    reads = list(synth.make_read_set(1, i, **synth.CONFIGS[2]))
    cut = [reads[0]]
    for r in reads[1:]:
        a = int(rng.integers(0, len(r) // 10 + 1)); b = len(r) - int(rng.integers(0, len(r) // 10 + 1))
        cut.append(r[a:b])
    sets.append(cut)
params = api.Params(**WORKLOADS["cfg2"]["params"])
enc = api.EncodedSets(sets, params.m)
api.msa_batch(None, params, encoded=enc, n_threads=16)
t0 = time.perf_counter()
for _ in range(2):
    res = api.msa_batch(None, params, encoded=enc, n_threads=16)
dt = (time.perf_counter() - t0) / 2
print(json.dumps({"read_sets_per_s": round(n_sets / dt, 1), "ms": round(dt * 1e3, 2), "cells": sum(r.n_cells for r in res), "timing": api.msa_timing()}), flush=True)
