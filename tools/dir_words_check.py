"""GPU debug tool: the direction words the row loops wrote (flat API, trace mode with ABPOA_HIP_DIRTRACE=1) against the words oracle/dir_model.c derives
from the oracle's scores, cell by cell, on normalised fields (kM, kE, H == Ein, E opened, H == F).  usage: python tools/dir_words_check.py <golden label prefix> ..."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ABPOA_HIP_DIRTRACE"] = "1"
import numpy as np
import helpers as H
from abpoa_amd import ffi

def model_words(case):
    lib = H.oracle_lib(); case.reset()
    res, tr = H.Result(), H.OracleTrace()
    assert lib.abpoa_oracle_align(C.byref(case.sc), C.byref(case.pb), C.byref(res), C.byref(tr)) == 0
    lib.abpoa_oracle_dir_words.argtypes = [C.POINTER(H.Scoring), C.POINTER(H.Problem), C.POINTER(H.OracleTrace), C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    w = np.zeros((case.n_rows, tr.width), np.uint32)
    rc = lib.abpoa_oracle_dir_words(C.byref(case.sc), C.byref(case.pb), C.byref(tr), res.best_row, res.best_col, w.ctypes.data_as(C.POINTER(C.c_uint32)))
    beg = np.ctypeslib.as_array(tr.dp_beg, (case.n_rows,)).copy(); end = np.ctypeslib.as_array(tr.dp_end, (case.n_rows,)).copy()
    lib.abpoa_oracle_free_trace(C.byref(tr)); H._libc.free(C.cast(res.cigar, C.c_void_p))
    return rc, w, beg, end

def norm(w, convex, o1, o2):
    if not convex:
        return np.stack([w & 15, (w >> 4) & 15, ((w >> 8) & 7) == o1, ((w >> 8) & 7) == 0, ((w >> 11) & 7) == 0], -1).astype(np.int32)
    return np.stack([w & 15, (w >> 4) & 15, (w >> 8) & 15, ((w >> 12) & 7) == o1, ((w >> 12) & 7) == 0, ((w >> 15) & 31) == o2, ((w >> 15) & 31) == 0,
                     ((w >> 20) & 7) == 0, ((w >> 23) & 31) == 0], -1).astype(np.int32)

lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
for label, path in H.golden_cases():
    if not any(label.startswith(a) for a in sys.argv[1:]):
        continue
    g = H.read_abpg(path); case = H.FlatCase(g)
    rc, mw, beg, end = model_words(case)
    if rc != 0:
        print(label, "plane does not apply"); continue
    h = H.run_hip([case], want_trace=True)[0]
    convex = int(g["gap_mode"][0]) == 2; o1, o2 = int(g["gap_open1"][0]), int(g["gap_open2"][0])
    pn = 16 if h.bits == 16 else 8; P = h.P
    bad_rows = []
    for r in range(1, case.n_rows - 1):
        if h.dp_beg_sn[r] < 0: continue
        W = (h.dp_end_sn[r] - h.dp_beg_sn[r] + 1) * pn; o = int(h.row_off[r])
        gw = h.planes[o:o + W].astype(np.int64) & (0xffff if h.bits == 16 else 0xffffffff)
        if convex and h.bits == 16: gw = gw | ((h.planes[o + W:o + 2 * W].astype(np.int64) & 0xffff) << 16)
        m = mw[r, h.dp_beg_sn[r] * pn: h.dp_beg_sn[r] * pn + W].astype(np.int64)
        # columns past the query are never read
        ncol = min(W, case.qlen + 1 - h.dp_beg_sn[r] * pn)
        a, b = norm(gw[:ncol], convex, o1, o2), norm(m[:ncol], convex, o1, o2)
        d = np.nonzero((a != b).any(-1))[0]
        if len(d): bad_rows.append((r, len(d), int(d[0]), hex(int(gw[d[0]])), hex(int(m[d[0]])), int(case.pred_off[r + 1] - case.pred_off[r]), W))
    print(label, "rows", case.n_rows, "rows with differing cells:", len(bad_rows), "first:", bad_rows[:8])
