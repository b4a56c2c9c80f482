# Row census of the narrow loop WITH the assembly loop (make -C abpoa_amd/csrc census -> libabpoa_hip_cnt.so): rows and clock ticks per alignment in the assembly
# loop, in the C++ copies of the straight-line body, in the all-chunks / exact bodies and in the tile switches, for the last rounds of 256 configs[1] read-sets.
# usage (GPU box): bash tools/asm_census.sh > gpurun_out/r5_asm_census.txt
[ -f abpoa_amd/libabpoa_hip_cnt.so ] || make -C abpoa_amd/csrc census > /dev/null 2>&1      # (the diagnostic library does not travel: built on the box, ~2 min)
export ABPOA_HIP_LIB=$PWD/abpoa_amd/libabpoa_hip_cnt.so ABPOA_HIP_LOCKSTEP=1 ABPOA_HIP_DBG=128 ABPOA_HIP_ROW_CENSUS=2 ABPOA_HIP_DEVSYNC=1 ABPOA_HIP_IMBAL=1
python3 - <<'PY' 2>&1 | grep -E "census|slowest row loop" | tail -8
import sys
sys.path.insert(0, ".")
from abpoa_amd import api, ffi, synth, workloads
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
w = workloads.WORKLOADS["cfg2"]
sets = [synth.make_read_set(1, i, **synth.CONFIGS[2]) for i in range(256)]
res = api.msa_batch(sets, api.Params(**w["params"]), n_threads=8)
assert all(r.status == 0 for r in res)
PY
