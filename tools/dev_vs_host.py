"""Device-resident driver vs host driver on synthetic read-sets: consensus, coverage and cell counts must be identical.
usage: dev_vs_host.py [n_sets] [n_reads] [len] [err]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abpoa_amd import api, ffi, synth
n_sets = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n_reads = int(sys.argv[2]) if len(sys.argv) > 2 else 12
ln = int(sys.argv[3]) if len(sys.argv) > 3 else 300
err = float(sys.argv[4]) if len(sys.argv) > 4 else 0.05
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
sets = [synth.make_read_set(7, i, n_reads, ln, err) for i in range(n_sets)]
p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
os.environ["ABPOA_HIP_HOSTGRAPH"] = "1"
t = time.time(); host = api.msa_batch(sets, p, n_threads=8); th = time.time() - t
os.environ["ABPOA_HIP_HOSTGRAPH"] = "0"
t = time.time(); dev = api.msa_batch(sets, p, n_threads=8); td = time.time() - t
tm = api.msa_timing()
bad = 0
for i, (a, b) in enumerate(zip(dev, host)):
    if a.status != 0 or b.status != 0 or a.cons_seq != b.cons_seq or a.cons_cov != b.cons_cov or a.n_cells != b.n_cells:
        bad += 1
        if bad <= 5: print("set", i, "status", a.status, b.status, "len", len(a.cons_seq), len(b.cons_seq), "cells", a.n_cells, b.n_cells, "same seq", a.cons_seq == b.cons_seq, "same cov", a.cons_cov == b.cons_cov)
print(f"{n_sets} sets x {n_reads} x {ln}: mismatching sets {bad}; host {th:.3f}s device {td:.3f}s; timing {tm}")
sys.exit(1 if bad else 0)
