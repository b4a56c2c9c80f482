import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers as H
from abpoa_amd import ffi
name = sys.argv[1]; rows = [int(x) for x in sys.argv[2].split(",")]
path = [p for l, p in H.golden_cases() if l == name][0]
g = H.read_abpg(path); case = H.FlatCase(g)
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
o = H.run_oracle(case); h = H.run_hip([case])[0]
pn = 16 if o.bits == 16 else 8; P = o.P
for r in rows:
    a, b = int(o.row_off[r]), int(o.row_off[r + 1]); wv = (b - a) // P
    print("row", r, "band", o.dp_beg[r], o.dp_end[r], "hip band", h.dp_beg[r], h.dp_end[r], "preds", case.pred_row[case.pred_off[r]:case.pred_off[r+1]], "maxi", o.row_max_i[r], h.row_max_i[r])
    for p in range(P):
        ov, hv = o.planes[a + p * wv:a + (p + 1) * wv], h.planes[a + p * wv:a + (p + 1) * wv]
        bad = np.nonzero(ov != hv)[0]
        print(" plane", p, "ndiff", len(bad), "first bad cols", (bad[:8] + o.dp_beg[r]).tolist())
        if len(bad): print("   oracle", ov[max(0,bad[0]-3):bad[0]+6].tolist(), "\n   hip   ", hv[max(0,bad[0]-3):bad[0]+6].tolist())
