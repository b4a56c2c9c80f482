import ctypes as C, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from abpoa_amd import api, ffi, synth
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
sets = [synth.make_read_set(1, i, **synth.CONFIGS[2]) for i in range(int(sys.argv[1]) if len(sys.argv)>1 else 200)]
enc = api.EncodedSets(sets, 5)
api.msa_batch(None, p, encoded=enc, n_threads=32)
lib.abpoa_hip_reset_stats(); d=(C.c_longlong*10)(); lib.abpoa_hip__debug_clocks(d)
t=time.time(); api.msa_batch(None, p, encoded=enc, n_threads=32); dt=time.time()-t
lib.abpoa_hip__debug_clocks(d); st=ffi.stats()
print('wall %.3f kernel_ms %.1f launches %d'%(dt, st['kernel_ms'], st['n_launches']))
print('dp ticks/row %.0f  bt ticks/step %.0f  rows %d steps %d  dp_total %.3g bt_total %.3g'%(d[0]/max(1,d[2]), d[1]/max(1,d[3]), d[2], d[3], d[0], d[1]))
print('segments ticks/row: hdr %.0f gather %.0f fchain %.0f store %.0f argmax+outs %.0f top %.0f' % tuple(d[4+i]/max(1,d[2]) for i in range(6)))
print(api.msa_timing())
