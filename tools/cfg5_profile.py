import sys, time, os
wt = os.environ.get("WITH_TORCH", "")
if wt:
    import torch
    if "t1" in wt: torch.set_num_threads(1)
    if "cuda" in wt: torch.cuda.set_device(0); torch.cuda.synchronize()
sys.path.insert(0, "/root/repo")
from abpoa_amd import api, ffi, synth, workloads as W
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
w = W.WORKLOADS["cfg5"]
sets = [synth.make_read_set(1, i, **synth.CONFIGS[5]) for i in range(1000)]
p = api.Params(**w["params"]); enc = api.EncodedSets(sets, p.m)
for it in range(3):
    lib.abpoa_hip_reset_stats(); t0 = time.time()
    res = api.msa_batch(None, p, out_cons=False, out_msa=True, encoded=enc, n_threads=int(os.environ.get("NT", "16")))
    dt = time.time() - t0
    print(it, round(dt, 3), {k: round(v, 1) if isinstance(v, float) else v for k, v in ffi.stats().items()}, {k: round(v, 3) if isinstance(v, float) else v for k, v in api.msa_timing().items()}, flush=True)
