"""Debug aid: first row where the HIP trace differs from the oracle (band, arg-max, plane cells).  usage: dbg_first_diff.py case [case...]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, helpers as H
from abpoa_amd import ffi
lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
for name in sys.argv[1:]:
    path = [p for l, p in H.golden_cases() if l == name][0]
    g = H.read_abpg(path); case = H.FlatCase(g)
    o = H.run_oracle(case); h = H.run_hip([case])[0]
    pn = 16 if o.bits == 16 else 8; P = o.P
    print(name, "status", h.status, "bits", o.bits, "rows", case.n_rows, "score", h.best_score, o.best_score)
    nbad = 0
    for r in range(1, case.n_rows - 1):
        msg = []
        if h.dp_beg_sn[r] != o.dp_beg_sn[r] or h.dp_end_sn[r] != o.dp_end_sn[r]: msg.append(f"band hip {h.dp_beg_sn[r]},{h.dp_end_sn[r]} vs {o.dp_beg_sn[r]},{o.dp_end_sn[r]}")
        if h.row_max_i[r] != o.row_max_i[r]: msg.append(f"maxi hip {h.row_max_i[r]} vs {o.row_max_i[r]}")
        if True:
            a, b = int(o.row_off[r]), int(o.row_off[r + 1]); wv = (b - a) // P
            ha = int(h.row_off[r])
            for p in range(P):
                ov, hv = o.planes[a + p * wv:a + (p + 1) * wv], h.planes[ha + p * wv:ha + (p + 1) * wv]
                bad = np.nonzero(ov != hv)[0]
                if len(bad): msg.append(f"plane {p} ndiff {len(bad)} first col {bad[0] + o.dp_beg[r]} (rel {bad[0]}) oracle {ov[max(0,bad[0]-2):bad[0]+5].tolist()} hip {hv[max(0,bad[0]-2):bad[0]+5].tolist()}")
        if msg:
            ps = case.pred_row[case.pred_off[r]:case.pred_off[r+1]].tolist()
            print(" row", r, "preds", ps, "pred bands", [(int(o.dp_beg_sn[p]), int(o.dp_end_sn[p])) for p in ps], "band", int(o.dp_beg_sn[r]), int(o.dp_end_sn[r]), "|", "; ".join(msg))
            nbad += 1
            if nbad >= 6: break
    if nbad == 0: print(" rows identical; left/right equal:", np.array_equal(h.left, o.left), np.array_equal(h.right, o.right), "cigar equal:", np.array_equal(h.cigar, o.cigar))
