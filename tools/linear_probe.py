"""Linear-gap job (256 x 50 x 1 kb) on the device-resident driver with the phase breakdown.  usage: [ABPOA_HIP_VERBOSE=1] python tools/linear_probe.py [n_sets] [affine|linear|unbanded]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from abpoa_amd import api, ffi, synth
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mode = sys.argv[2] if len(sys.argv) > 2 else "linear"
kw = dict(linear=dict(gap_open1=0, gap_open2=0, gap_ext1=2), affine=dict(gap_open1=4, gap_open2=0, gap_ext1=2), unbanded=dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=-1))[mode]
sets = [synth.make_read_set(5, i, 50, 1000, 0.05) for i in range(n)]
p = api.Params(**kw)
api.msa_batch(sets[:32], p, n_threads=16)
for _ in range(2):
    t = time.time(); r = api.msa_batch(sets, p, n_threads=16); dt = time.time() - t
    print(f"{mode} {n} sets: {n / dt:.1f} sets/s  ok {all(x.status == 0 for x in r)}", flush=True)
