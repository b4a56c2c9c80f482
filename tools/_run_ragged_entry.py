import sys, os, json
sys.path.insert(0, os.getcwd())
import bench
from abpoa_amd import ffi
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
print(json.dumps(bench.ragged_entry(16)))
