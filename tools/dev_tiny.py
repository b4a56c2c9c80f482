import os, sys
sys.path.insert(0, os.getcwd())
from abpoa_amd import api, ffi, synth
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
sets = [synth.make_read_set(7, i, 3, 40, 0.05) for i in range(2)]
p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
print("device run", flush=True)
dev = api.msa_batch(sets, p, n_threads=2)
print([d.cons_seq for d in dev], api.msa_timing(), flush=True)
lib.abpoa_hip_reset_stats()
dev = api.msa_batch(sets, p, n_threads=2)
print("stats after ONE call:", ffi.stats(), api.msa_timing())
