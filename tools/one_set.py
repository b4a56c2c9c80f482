# One bench read-set (set 2 of a workload) through the read-set API, digest checked: python tools/one_set.py cfg4   (diagnostic; honours the ABPOA_HIP_* switches)
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from abpoa_amd import api, synth, workloads as W, ffi
ffi.check(ffi.lib().abpoa_hip_init(0))
wl = sys.argv[1]
w = W.WORKLOADS[wl]
r = api.msa_batch([synth.make_read_set(1, 2, **synth.CONFIGS[w["cfg"]])], api.Params(**w["params"]))[0]
print(wl, r.status, W.output_sha(api.format_output(r)) == W.load_digests(wl)[2])
