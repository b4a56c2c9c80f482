"""Shared test plumbing: .abpg golden-container reader, ctypes views of include/abpoa_hip.h structs,
the oracle binding, and comparison helpers.  Nothing here is imported by the product package."""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
REF_DIR = os.path.join(ORACLE_DIR, "_ref")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")
REFERENCE_TREE = "/root/reference"
sys.path.insert(0, ROOT)

_DT = {0: np.uint8, 1: np.int32, 2: np.int64, 3: np.float32, 4: np.int16, 5: np.uint64}


def read_abpg(path):
    """Parse a container written by oracle/ref_dump.c -> dict name -> np.ndarray (scalars as 1-elem arrays)."""
    if path.endswith(".gz"):
        import gzip
        raw = gzip.open(path, "rb").read()
    else:
        raw = open(path, "rb").read()
    assert raw[:8] == b"ABPG0001", path
    off, out = 8, {}
    while off < len(raw):
        name = raw[off:off + 24].split(b"\0", 1)[0].decode()
        dt, _pad = np.frombuffer(raw, np.int32, 2, off + 24)
        cnt = int(np.frombuffer(raw, np.int64, 1, off + 32)[0])
        off += 40
        dtype = np.dtype(_DT[int(dt)])
        out[name] = np.frombuffer(raw, dtype, cnt, off).copy()
        off += cnt * dtype.itemsize
    return out


from abpoa_amd.ffi import Scoring, Problem, Result, Trace as HipTrace  # noqa: E402  (C-ABI structs)


class OracleTrace(C.Structure):
    _fields_ = [("bits", C.c_int32), ("pn", C.c_int32), ("n_planes", C.c_int32), ("dp_sn", C.c_int32),
                ("width", C.c_int32), ("inf_min", C.c_int32), ("n_rows", C.c_int32),
                ("dp_beg", C.POINTER(C.c_int32)), ("dp_end", C.POINTER(C.c_int32)),
                ("dp_beg_sn", C.POINTER(C.c_int32)), ("dp_end_sn", C.POINTER(C.c_int32)),
                ("row_max_i", C.POINTER(C.c_int32)), ("planes", C.POINTER(C.c_int32))]


def _p(arr, ctype):
    return arr.ctypes.data_as(C.POINTER(ctype))


class FlatCase:
    """A (scoring, problem) pair backed by numpy arrays (kept alive on self)."""

    def __init__(self, d):
        """d: dict as produced by read_abpg (input section) or built by hand with the same keys."""
        g = lambda k: int(np.asarray(d[k]).reshape(-1)[0])
        self.d = d
        self.mat = np.ascontiguousarray(d["mat"], np.int32)
        self.query = np.ascontiguousarray(d["query"], np.uint8)
        self.row_base = np.ascontiguousarray(d["row_base"], np.uint8)
        self.row_node_id = np.ascontiguousarray(d["row_node_id"], np.int32)
        self.row_remain = np.ascontiguousarray(d["row_remain"], np.int32)
        self.row_active = np.ascontiguousarray(d["row_active"], np.uint8)
        self.pred_off = np.ascontiguousarray(d["pred_off"], np.int32)
        self.pred_row = np.ascontiguousarray(d["pred_row"], np.int32)
        self.out_off = np.ascontiguousarray(d["out_off"], np.int32)
        self.out_row = np.ascontiguousarray(d["out_row"], np.int32)
        self.left = np.ascontiguousarray(d["left_in"], np.int32).copy()
        self.right = np.ascontiguousarray(d["right_in"], np.int32).copy()
        if self.query.size == 0:
            self.query = np.zeros(1, np.uint8)
        if self.pred_row.size == 0:
            self.pred_row = np.zeros(1, np.int32)
        if self.out_row.size == 0:
            self.out_row = np.zeros(1, np.int32)
        self.sc = Scoring(g("m"), _p(self.mat, C.c_int32), g("max_mat"), g("min_mis"), g("gap_open1"), g("gap_ext1"),
                          g("gap_open2"), g("gap_ext2"), g("align_mode"), g("gap_mode"), g("wb"),
                          float(np.asarray(d["wf"], np.float32).reshape(-1)[0]), g("zdrop"), g("ret_cigar"), g("rev_cigar"))
        self.n_rows, self.qlen = g("n_rows"), g("qlen")
        self.reset()

    def reset(self):
        self.left[:] = self.d["left_in"]
        self.right[:] = self.d["right_in"]
        self.pb = Problem(self.n_rows, self.qlen, _p(self.query, C.c_uint8), _p(self.row_base, C.c_uint8),
                          _p(self.row_node_id, C.c_int32), _p(self.row_remain, C.c_int32), _p(self.row_active, C.c_uint8),
                          _p(self.pred_off, C.c_int32), _p(self.pred_row, C.c_int32), _p(self.out_off, C.c_int32),
                          _p(self.out_row, C.c_int32), _p(self.left, C.c_int32), _p(self.right, C.c_int32))


_oracle = None


def oracle_lib():
    """Build (if needed) and load oracle/liboracle_dp.so."""
    global _oracle
    if _oracle is None:
        so = os.path.join(ORACLE_DIR, "liboracle_dp.so")
        srcs = [os.path.join(ORACLE_DIR, f) for f in ("abpoa_dp_oracle.c", "abpoa_dp_oracle.h", "dir_model.c")] + [os.path.join(ROOT, "abpoa_amd", "csrc", "dir_plane.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
            subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "oracle_dp"])
        lib = C.CDLL(so)
        lib.abpoa_oracle_align.argtypes = [C.POINTER(Scoring), C.POINTER(Problem), C.POINTER(Result), C.POINTER(OracleTrace)]
        lib.abpoa_oracle_align.restype = C.c_int
        lib.abpoa_oracle_free_trace.argtypes = [C.POINTER(OracleTrace)]
        lib.abpoa_oracle_score_bits.argtypes = [C.POINTER(Scoring), C.c_int, C.c_int, C.POINTER(C.c_int32)]
        lib.abpoa_oracle_score_bits.restype = C.c_int
        lib.abpoa_oracle_dir_walk.argtypes = [C.POINTER(Scoring), C.POINTER(Problem), C.POINTER(OracleTrace), C.c_int, C.c_int, C.POINTER(Result), C.POINTER(C.c_int64)]
        lib.abpoa_oracle_dir_walk.restype = C.c_int
        _oracle = lib
    return _oracle


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


class OracleOut:
    pass


def run_oracle(case, want_trace=True):
    """Run the C oracle on a FlatCase; returns OracleOut with numpy copies of everything."""
    lib = oracle_lib()
    case.reset()
    res, tr = Result(), OracleTrace()
    rc = lib.abpoa_oracle_align(C.byref(case.sc), C.byref(case.pb), C.byref(res), C.byref(tr) if want_trace else None)
    o = OracleOut()
    o.rc, o.status, o.bits = rc, res.status, res.bits
    for k in ("best_score", "best_row", "best_col", "node_s", "node_e", "query_s", "query_e", "n_aln_bases",
              "n_matched_bases", "n_cigar", "n_cells"):
        setattr(o, k, getattr(res, k))
    o.cigar = np.ctypeslib.as_array(res.cigar, (res.n_cigar,)).copy() if res.n_cigar > 0 else np.zeros(0, np.uint64)
    if res.cigar:
        _libc.free(C.cast(res.cigar, C.c_void_p))
    o.left, o.right = case.left.copy(), case.right.copy()
    if want_trace:
        n = case.n_rows
        o.pn, o.P, o.width, o.inf_min = tr.pn, tr.n_planes, tr.width, tr.inf_min
        o.dp_beg = np.ctypeslib.as_array(tr.dp_beg, (n,)).copy()
        o.dp_end = np.ctypeslib.as_array(tr.dp_end, (n,)).copy()
        o.dp_beg_sn = np.ctypeslib.as_array(tr.dp_beg_sn, (n,)).copy()
        o.dp_end_sn = np.ctypeslib.as_array(tr.dp_end_sn, (n,)).copy()
        o.row_max_i = np.ctypeslib.as_array(tr.row_max_i, (n,)).copy()
        planes = np.ctypeslib.as_array(tr.planes, (n, tr.n_planes, tr.width))
        # compact to the band-compacted layout of the goldens / HIP trace
        act = o.dp_beg_sn >= 0
        o.row_off = np.zeros(n + 1, np.int64)
        chunks = []
        for r in range(n):
            o.row_off[r + 1] = o.row_off[r]
            if not act[r]:
                continue
            a, b = o.dp_beg_sn[r] * tr.pn, (o.dp_end_sn[r] + 1) * tr.pn
            chunks.append(planes[r, :, a:b].reshape(-1))
            o.row_off[r + 1] += tr.n_planes * (b - a)
        o.planes = np.concatenate(chunks) if chunks else np.zeros(0, np.int32)
        o.active = act
        lib.abpoa_oracle_free_trace(C.byref(tr))
    return o


def run_dir_model(case):
    """oracle/dir_model.c on a FlatCase: the oracle's alignment (trace kept), then the direction plane built from that trace and walked.
    Returns (oracle result fields, model result fields, stats[6]); rc_model == ABPOA_HIP_EINVAL where the plane does not apply."""
    lib = oracle_lib()
    case.reset()
    res, tr = Result(), OracleTrace()
    rc = lib.abpoa_oracle_align(C.byref(case.sc), C.byref(case.pb), C.byref(res), C.byref(tr))
    assert rc == 0 and res.status == 0
    FIELDS = ("best_row", "best_col", "node_s", "node_e", "query_s", "query_e", "n_aln_bases", "n_matched_bases", "n_cigar")

    def grab(r):
        d = {k: getattr(r, k) for k in FIELDS}
        d["cigar"] = np.ctypeslib.as_array(r.cigar, (r.n_cigar,)).copy() if r.n_cigar > 0 else np.zeros(0, np.uint64)
        if r.cigar:
            _libc.free(C.cast(r.cigar, C.c_void_p))
        return d
    res2, stats = Result(), (C.c_int64 * 10)()
    rc2 = lib.abpoa_oracle_dir_walk(C.byref(case.sc), C.byref(case.pb), C.byref(tr), res.best_row, res.best_col, C.byref(res2), stats)
    a, b = grab(res), (grab(res2) if rc2 == 0 else None)
    lib.abpoa_oracle_free_trace(C.byref(tr))
    return rc2, a, b, list(stats)


def dir_words_mismatches(case, g):
    """GPU: the direction words the row loops wrote (flat API in trace mode with ABPOA_HIP_DIRTRACE=1: plane 0 of the trace carries the words) against the
    words oracle/dir_model.c derives from the oracle's scores, cell by cell, on the fields a walk decides with (kM, kE, H == Ein, E opened from H, H == F;
    where F came from is compared through the walk itself: the cigar).  Returns None when the plane does not apply, else the list of differing rows."""
    lib = oracle_lib(); case.reset()
    res, tr = Result(), OracleTrace()
    assert lib.abpoa_oracle_align(C.byref(case.sc), C.byref(case.pb), C.byref(res), C.byref(tr)) == 0
    lib.abpoa_oracle_dir_words.argtypes = [C.POINTER(Scoring), C.POINTER(Problem), C.POINTER(OracleTrace), C.c_int, C.c_int, C.POINTER(C.c_uint32)]
    mw = np.zeros((case.n_rows, tr.width), np.uint32)
    rc = lib.abpoa_oracle_dir_words(C.byref(case.sc), C.byref(case.pb), C.byref(tr), res.best_row, res.best_col, mw.ctypes.data_as(C.POINTER(C.c_uint32)))
    lib.abpoa_oracle_free_trace(C.byref(tr)); _libc.free(C.cast(res.cigar, C.c_void_p))
    if rc != 0:
        return None
    old = os.environ.get("ABPOA_HIP_DIRTRACE")
    os.environ["ABPOA_HIP_DIRTRACE"] = "1"
    try:
        h = run_hip([case], want_trace=True)[0]
    finally:
        if old is None:
            del os.environ["ABPOA_HIP_DIRTRACE"]
        else:
            os.environ["ABPOA_HIP_DIRTRACE"] = old
    convex = int(g["gap_mode"][0]) == 2
    o1, o2 = int(g["gap_open1"][0]), int(g["gap_open2"][0])

    def norm(w):
        if not convex:
            return np.stack([w & 15, (w >> 4) & 15, ((w >> 8) & 7) == o1, ((w >> 8) & 7) == 0, ((w >> 11) & 7) == 0], -1).astype(np.int32)
        return np.stack([w & 15, (w >> 4) & 15, (w >> 8) & 15, ((w >> 12) & 7) == o1, ((w >> 12) & 7) == 0, ((w >> 15) & 31) == o2, ((w >> 15) & 31) == 0,
                         ((w >> 20) & 7) == 0, ((w >> 23) & 31) == 0], -1).astype(np.int32)
    pn = 16 if h.bits == 16 else 8
    bad = []
    for r in range(1, case.n_rows - 1):
        if h.dp_beg_sn[r] < 0:
            continue
        W = (h.dp_end_sn[r] - h.dp_beg_sn[r] + 1) * pn
        o = int(h.row_off[r])
        gw = h.planes[o:o + W].astype(np.int64) & (0xffff if h.bits == 16 else 0xffffffff)
        if convex and h.bits == 16:
            gw = gw | ((h.planes[o + W:o + 2 * W].astype(np.int64) & 0xffff) << 16)
        m = mw[r, h.dp_beg_sn[r] * pn: h.dp_beg_sn[r] * pn + W].astype(np.int64)
        ncol = min(W, case.qlen + 1 - h.dp_beg_sn[r] * pn)          # (columns past the query are never read)
        d = np.nonzero((norm(gw[:ncol]) != norm(m[:ncol])).any(-1))[0]
        if len(d):
            bad.append((r, len(d), int(d[0]), hex(int(gw[d[0]])), hex(int(m[d[0]]))))
    return bad


def mix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def row_checksums(planes, row_off, dp_beg_sn, dp_end_sn, pn, P):
    """Same weighted sum as oracle/ref_dump.c (row_checksum record)."""
    n = len(dp_beg_sn)
    out = np.zeros(n, np.uint64)
    with np.errstate(over="ignore"):
        for r in range(n):
            if dp_beg_sn[r] < 0 or row_off[r + 1] == row_off[r]:
                continue
            wv = (int(dp_end_sn[r]) - int(dp_beg_sn[r]) + 1) * pn
            cells = planes[row_off[r]:row_off[r] + P * wv].astype(np.int64).astype(np.uint32).astype(np.uint64).reshape(P, wv)
            cols = (np.arange(wv, dtype=np.uint64) + np.uint64(int(dp_beg_sn[r]) * pn))
            wgt = mix64(cols[None, :] * np.uint64(8) + np.arange(P, dtype=np.uint64)[:, None])
            out[r] = np.sum(cells * wgt, dtype=np.uint64)
    return out


def have_ref():
    return os.path.exists(os.path.join(REF_DIR, "ref_dump"))


def run_ref_dump(fasta, outdir, opts, reads="all", planes=1, sub=None):
    os.makedirs(outdir, exist_ok=True)
    cmd = [os.path.join(REF_DIR, "ref_dump")] + list(opts) + ["-D", outdir, "-R", reads, "-P", str(planes)]
    if sub:
        cmd += ["-G", "%d,%d" % sub]
    subprocess.check_call(cmd + [fasta], stderr=subprocess.DEVNULL)


def compare_with_golden(o, g, check_planes=True, label=""):
    """Assert that an OracleOut-like object `o` (oracle or HIP) equals golden dict `g` bit for bit."""
    assert o.status == 0, f"{label}: status {o.status}"
    assert o.bits == int(g["bits"][0]), f"{label}: bits {o.bits} vs {int(g['bits'][0])}"
    assert o.best_score == int(g["best_score"][0]), f"{label}: best_score {o.best_score} vs {int(g['best_score'][0])}"
    gc = g["cigar"]
    assert o.n_cigar == len(gc) and np.array_equal(o.cigar, gc), f"{label}: cigar differs (n {o.n_cigar} vs {len(gc)})"
    for k in ("node_s", "node_e", "query_s", "query_e", "n_aln_bases", "n_matched_bases"):
        if len(gc) > 0:
            assert getattr(o, k) == int(g[k][0]), f"{label}: {k} {getattr(o, k)} vs {int(g[k][0])}"
    assert o.n_cells == int(g["n_cells"][0]), f"{label}: n_cells {o.n_cells} vs {int(g['n_cells'][0])}"
    if int(g["wb"][0]) >= 0:
        assert np.array_equal(o.left, g["left_out"]), f"{label}: max_pos_left differs"
        assert np.array_equal(o.right, g["right_out"]), f"{label}: max_pos_right differs"
    if hasattr(o, "dp_beg"):
        m = g["dp_beg_sn"] >= 0
        for k in ("dp_beg", "dp_end", "dp_beg_sn", "dp_end_sn"):
            a, b = getattr(o, k)[m], g[k][m]
            if not np.array_equal(a, b):
                bad = np.nonzero(a != b)[0][0]
                raise AssertionError(f"{label}: {k} differs first at active row #{bad}: {a[bad]} vs {b[bad]}")
        assert np.array_equal(o.row_off, g["row_off"]), f"{label}: row_off differs"
        if check_planes:
            pn = 16 if o.bits == 16 else 8
            P = int(g["n_planes"][0])
            if "planes" in g:
                a, b = o.planes.astype(np.int64), g["planes"].astype(np.int64)
                if not np.array_equal(a, b):
                    bad = int(np.nonzero(a != b)[0][0])
                    r = int(np.searchsorted(g["row_off"], bad, side="right") - 1)
                    wv = (int(g["dp_end_sn"][r]) - int(g["dp_beg_sn"][r]) + 1) * pn
                    rel = bad - int(g["row_off"][r])
                    raise AssertionError(f"{label}: plane cell differs: row {r} plane {rel // wv} col "
                                         f"{int(g['dp_beg_sn'][r]) * pn + rel % wv}: {a[bad]} vs {b[bad]}")
            cs = row_checksums(o.planes, o.row_off, g["dp_beg_sn"], g["dp_end_sn"], pn, P)
            assert np.array_equal(cs, g["row_checksum"]), f"{label}: row checksum differs at rows {np.nonzero(cs != g['row_checksum'])[0][:5]}"


def golden_cases(with_planes_only=False):
    """[(label, path)] of every committed per-alignment golden container."""
    import glob
    out = []
    for f in sorted(glob.glob(os.path.join(GOLDEN_DIR, "*", "aln_*.abpg.gz"))):
        out.append((os.path.basename(os.path.dirname(f)) + "/" + os.path.basename(f)[:-8], f))
    return out


def run_hip(cases, want_trace=True):
    """Run FlatCases through the C-ABI (abpoa_hip_align_batch); returns a list of OracleOut-like objects."""
    from abpoa_amd import ffi
    lib = ffi.lib()
    n = len(cases)
    for c in cases:
        c.reset()
        assert c.sc.m == cases[0].sc.m
    pbs = (Problem * n)(*[c.pb for c in cases])
    res = (Result * n)()
    rc = lib.abpoa_hip_align_batch(C.byref(cases[0].sc), n, pbs, res, ffi.FLAG_TRACE if want_trace else 0)
    ffi.check(rc)
    outs = []
    for i, case in enumerate(cases):
        r = res[i]
        o = OracleOut()
        o.rc, o.status, o.bits = rc, r.status, r.bits
        for k in ("best_score", "best_row", "best_col", "node_s", "node_e", "query_s", "query_e", "n_aln_bases",
                  "n_matched_bases", "n_cigar", "n_cells"):
            setattr(o, k, getattr(r, k))
        o.cigar = np.ctypeslib.as_array(r.cigar, (r.n_cigar,)).copy() if r.n_cigar > 0 else np.zeros(0, np.uint64)
        o.left, o.right = case.left.copy(), case.right.copy()
        if want_trace and r.trace:
            t = r.trace.contents
            nr = case.n_rows
            o.P = t.n_planes
            for k in ("dp_beg", "dp_end", "dp_beg_sn", "dp_end_sn", "row_max_i"):
                setattr(o, k, np.ctypeslib.as_array(getattr(t, k), (nr,)).copy())
            o.row_off = np.ctypeslib.as_array(t.row_off, (nr + 1,)).copy()
            tot = int(o.row_off[-1])
            dt = np.int16 if t.bits == 16 else np.int32
            o.planes = np.frombuffer((C.c_char * (tot * dt().itemsize)).from_address(t.planes), dt).astype(np.int32) if tot else np.zeros(0, np.int32)
        lib.abpoa_hip_free_result(C.byref(res[i]))
        outs.append(o)
    return outs


def compare_outs(a, b, label=""):
    """HIP vs oracle on the same problem (both OracleOut-like)."""
    assert a.status == b.status == 0, f"{label}: status {a.status}/{b.status}"
    for k in ("bits", "best_score", "best_row", "best_col", "n_cigar", "n_cells", "node_s", "node_e", "query_s",
              "query_e", "n_aln_bases", "n_matched_bases"):
        assert getattr(a, k) == getattr(b, k), f"{label}: {k} {getattr(a, k)} vs {getattr(b, k)}"
    assert np.array_equal(a.cigar, b.cigar), f"{label}: cigar differs"
    assert np.array_equal(a.left, b.left) and np.array_equal(a.right, b.right), f"{label}: band state differs"
    if hasattr(a, "dp_beg") and hasattr(b, "dp_beg"):
        for k in ("dp_beg", "dp_end", "dp_beg_sn", "dp_end_sn", "row_off"):
            assert np.array_equal(getattr(a, k), getattr(b, k)), f"{label}: {k} differs"
        m = b.dp_beg_sn >= 0
        m[0] = False
        assert np.array_equal(a.row_max_i[m], b.row_max_i[m]), f"{label}: row arg-max differs"
        if not np.array_equal(a.planes, b.planes):
            bad = int(np.nonzero(a.planes != b.planes)[0][0])
            r = int(np.searchsorted(b.row_off, bad, side="right") - 1)
            pn = 16 if b.bits == 16 else 8
            wv = (int(b.dp_end_sn[r]) - int(b.dp_beg_sn[r]) + 1) * pn
            rel = bad - int(b.row_off[r])
            raise AssertionError(f"{label}: plane cell differs: row {r} plane {rel // wv} col "
                                 f"{int(b.dp_beg_sn[r]) * pn + rel % wv}: {a.planes[bad]} vs {b.planes[bad]}")


_shim = None


def cpu_shim_lib():
    """CPU-only build of the product's host layer with an oracle-backed aligner (tests/cpu_shim.cpp)."""
    global _shim
    if _shim is None:
        bdir = os.path.join(ROOT, "tests", "_build")
        os.makedirs(bdir, exist_ok=True)
        so = os.path.join(bdir, "libcpu_shim.so")
        srcs = [os.path.join(ROOT, "tests", "cpu_shim.cpp"), os.path.join(ROOT, "abpoa_amd", "csrc", "poa_graph.cpp"),
                os.path.join(ROOT, "abpoa_amd", "csrc", "msa_batch.cpp"), os.path.join(ROOT, "abpoa_amd", "csrc", "engine_options.cpp")]
        deps = srcs + [os.path.join(ORACLE_DIR, "abpoa_dp_oracle.c"), os.path.join(ORACLE_DIR, "dir_model.c"), os.path.join(ORACLE_DIR, "abpoa_dp_oracle.h"), os.path.join(ROOT, "abpoa_amd", "csrc", "dir_plane.h"), os.path.join(ROOT, "abpoa_amd", "csrc", "poa_graph.h"), os.path.join(ROOT, "abpoa_amd", "csrc", "msa_batch.h"), os.path.join(ROOT, "abpoa_amd", "csrc", "batch_types.h"),
                       os.path.join(ROOT, "include", "abpoa_hip.h")]
        if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(d) for d in deps):
            objs = []
            for f in ("abpoa_dp_oracle", "dir_model"):
                objs.append(os.path.join(bdir, f + ".o"))
                subprocess.check_call(["gcc", "-O2", "-fPIC", "-c", "-I" + os.path.join(ROOT, "include"), "-o", objs[-1], os.path.join(ORACLE_DIR, f + ".c")])
            subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-I" + os.path.join(ROOT, "include"), "-o", so] + srcs + objs)
        _shim = C.CDLL(so)
        _shim.abpoa_shim_dir_checked.restype = C.c_longlong
    return _shim
