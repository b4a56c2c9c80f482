"""Device-resident driver vs host driver on the general kernel's jobs (linear gaps, extension mode, no band): read-sets/s of 256 x 50 x 1 kb sets.  usage: python tools/general_jobs_bench.py"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from abpoa_amd import api, ffi, synth
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
sets = [synth.make_read_set(5, i, 50, 1000, 0.05) for i in range(256)]
for name, kw in (("linear", dict(gap_open1=0, gap_open2=0, gap_ext1=2)), ("extend_convex", dict(aln_mode=2)), ("affine_unbanded", dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=-1)), ("affine_banded_fast", dict(gap_open1=4, gap_open2=0, gap_ext1=2))):
    p = api.Params(**kw)
    for host in (0, 1):
        os.environ["ABPOA_HIP_NO_DEVICE_GENERAL"] = str(host)
        os.environ["ABPOA_HIP_HOSTGRAPH"] = str(host) if name in ("affine_banded_fast", "linear") else "0"      # (jobs of the fast row loops: the host driver by its own switch)
        api.msa_batch(sets[:32], p, n_threads=16)
        t = time.time(); r = api.msa_batch(sets, p, n_threads=16); dt = time.time() - t
        tm = api.msa_timing()
        print(f"{name:20s} {'host driver' if host else 'device     '} {len(sets)/dt:8.1f} sets/s  n_host_sets {tm['n_host_sets']}  ok {all(x.status == 0 for x in r)}", flush=True)
        if host == 0: keep = [x.cons_seq for x in r]
        else: print("   same consensus:", keep == [x.cons_seq for x in r])
