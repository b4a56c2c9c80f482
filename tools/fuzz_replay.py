"""Replays one iteration of tools/fuzz_device_vs_oracle.py by its seed (the number the fuzzer prints), with the engine's pass log.  usage: python tools/fuzz_replay.py SEED"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ.setdefault("ABPOA_HIP_VERBOSE", "1")
import helpers as H                                   # noqa: E402
from abpoa_amd import api, ffi                        # noqa: E402
import fuzz_device_vs_oracle as F                     # noqa: E402
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
seed = int(sys.argv[1])
_orig = api.Params
def _p(**kw):
    print("params:", kw, flush=True); return _orig(**kw)
api.Params = _p
d, h, e, why = F.iteration(seed, H.cpu_shim_lib())
print(f"seed {seed}: {d} sets on the device, {h} through the host driver {why}, {e} with the same error status on both sides")
