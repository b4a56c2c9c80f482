"""Kernel-only micro-benchmark: one golden alignment replicated N times through the flat C-ABI.
usage: python tools/kernel_bench.py [case] [N] [ret_cigar]   (env ABPOA_HIP_DBG = ablation bits, timing only)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import helpers as H
from abpoa_amd import ffi
case_name = sys.argv[1] if len(sys.argv) > 1 else "s1k_ag_gb/aln_011"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
ret_cigar = int(sys.argv[3]) if len(sys.argv) > 3 else 1
path = [p for l, p in H.golden_cases() if l == case_name][0]
g = H.read_abpg(path)
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
cases = [H.FlatCase(g) for _ in range(N)]
for c in cases: c.sc.ret_cigar = ret_cigar
pbs = (ffi.Problem * N)(*[c.pb for c in cases]); res = (ffi.Result * N)()
d = (C.c_longlong * 10)()
for it in range(3):
    for c in cases: c.reset()
    for i, c in enumerate(cases): pbs[i] = c.pb
    lib.abpoa_hip_reset_stats(); lib.abpoa_hip__debug_clocks(d)
    ffi.check(lib.abpoa_hip_align_batch(C.byref(cases[0].sc), N, pbs, res, 0))
    lib.abpoa_hip__debug_clocks(d); st = ffi.stats()
    for i in range(N): lib.abpoa_hip_free_result(C.byref(res[i]))
rows = g["n_rows"][0]
print(f"{case_name} x{N} dbg={os.environ.get('ABPOA_HIP_DBG','0')} cigar={ret_cigar}: kernel {st['kernel_ms']:.2f} ms  rows {rows}  dp ticks/row {d[0]/max(1,d[2]):.0f}  bt ticks/step {d[1]/max(1,d[3]):.0f}  Gcells/s {st['n_cells']/st['kernel_ms']/1e6:.2f}")
if int(os.environ.get("ABPOA_HIP_DBG", "0")) & 256:
    print("   raw seg sums / N:", [hex(int(d[4 + i] // N)) for i in range(6)], "status of first:", res[0].status)
print("   segments ticks/row: " + "  ".join(f"s{i} {d[4+i]/max(1,d[2]):.0f}" for i in range(6)))
