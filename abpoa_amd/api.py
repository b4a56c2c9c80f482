"""Host-side mirror of the reference's parameter block and thin wrappers over the C-ABI.

`Params` mirrors abpoa_para_t (ref src/abpoa.h:62-81) with the defaults of abpoa_init_para
(src/abpoa_align.c:93-141) and the derived fields of abpoa_post_set_para (:143-168)."""
import ctypes as C
from collections.abc import Sequence

import numpy as np

from . import ffi, seqio

GLOBAL, LOCAL, EXTEND = 0, 1, 2
LINEAR, AFFINE, CONVEX = 0, 1, 2
OUT_CONS, OUT_MSA, AMB_STRAND = 1, 2, 4


class ReadSet(C.Structure):          # abpoa_hip_readset_t
    _fields_ = [("n_reads", C.c_int32), ("seqs", C.POINTER(C.POINTER(C.c_uint8))), ("lens", C.POINTER(C.c_int32)),
                ("weights", C.POINTER(C.POINTER(C.c_int32)))]


class Msa(C.Structure):              # abpoa_hip_msa_t
    _fields_ = [("status", C.c_int32), ("n_reads", C.c_int32), ("cons_len", C.c_int32),
                ("cons_base", C.POINTER(C.c_uint8)), ("cons_cov", C.POINTER(C.c_int32)), ("cons_node_id", C.POINTER(C.c_int32)),
                ("msa_len", C.c_int32), ("msa_rows", C.c_int32), ("msa_base", C.POINTER(C.c_uint8)), ("n_cells", C.c_int64),
                ("is_rc", C.POINTER(C.c_uint8))]


class MsaTiming(C.Structure):        # abpoa_hip_msa_timing_t
    _fields_ = [("host_sort_s", C.c_double), ("host_fuse_s", C.c_double), ("engine_s", C.c_double), ("cons_s", C.c_double),
                ("total_s", C.c_double), ("n_rounds", C.c_int32), ("n_threads", C.c_int32), ("n_groups", C.c_int32), ("n_host_sets", C.c_int32)]


class Params:
    def __init__(self, aln_mode=GLOBAL, is_aa=False, match=2, mismatch=4, score_matrix=None, gap_open1=4, gap_open2=24,
                 gap_ext1=2, gap_ext2=1, extra_b=10, extra_f=0.01, zdrop=-1):
        self.align_mode, self.m = aln_mode, (27 if is_aa else 5)
        self.match, self.mismatch, self.mat_fn = match, mismatch, score_matrix
        self.gap_open1, self.gap_open2, self.gap_ext1, self.gap_ext2 = gap_open1, gap_open2, gap_ext1, gap_ext2
        self.wb, self.wf, self.zdrop = extra_b, extra_f, zdrop
        self.finalize()

    def finalize(self):
        """abpoa_post_set_para: gap mode from the open penalties, LOCAL disables the band, matrix."""
        if self.gap_open1 == 0:
            self.gap_mode = LINEAR
        elif self.gap_open1 > 0 and self.gap_open2 == 0:
            self.gap_mode = AFFINE
        else:
            self.gap_mode = CONVEX
        if self.align_mode == LOCAL:
            self.wb = -1
        if self.mat_fn:
            self.mat, self.max_mat, self.min_mis = seqio.matrix_from_file(self.mat_fn, self.m)
        else:
            self.mat, self.max_mat, self.min_mis = seqio.simple_matrix(self.m, self.match, self.mismatch)
        self.mat = np.ascontiguousarray(self.mat, np.int32)
        return self

    def scoring(self):
        return ffi.Scoring(self.m, self.mat.ctypes.data_as(C.POINTER(C.c_int32)), self.max_mat, self.min_mis, self.gap_open1,
                           self.gap_ext1, self.gap_open2, self.gap_ext2, self.align_mode, self.gap_mode, self.wb,
                           float(self.wf), self.zdrop, 1, 0)


class _BatchOut:
    """Owns the result records of one abpoa_hip_msa_batch call; the library's arrays are released when the last SetResult goes."""
    __slots__ = ("lib", "out", "n")

    def __init__(self, lib, out, n):
        self.lib, self.out, self.n = lib, out, n

    def __del__(self):
        try:
            if hasattr(self.lib, "abpoa_hip_free_msa_array"):
                self.lib.abpoa_hip_free_msa_array(self.out, self.n)
            else:
                free_msa, out = self.lib.abpoa_hip_free_msa, self.out
                for i in range(self.n):
                    free_msa(C.byref(out[i]))
        except Exception:      # interpreter shutdown: the library may already be gone
            pass


class SetResult:
    """Result of one read-set: a view of the library's record.  Strings / lists are built on first access (building 1000 Python
    strings and coverage lists eagerly costs more than the whole GPU job of a batch)."""
    __slots__ = ("_o", "_owner", "_m", "_cons_seq", "_cov_list", "_msa_seq")

    def __init__(self, owner, rec, m):
        self._owner, self._o, self._m = owner, rec, m
        self._cons_seq = self._cov_list = self._msa_seq = None

    status = property(lambda self: self._o.status)
    is_rc = property(lambda self: [bool(self._o.is_rc[i]) for i in range(self._o.n_reads)] if self._o.is_rc else [False] * self._o.n_reads)
    n_cells = property(lambda self: self._o.n_cells)
    cons_len = property(lambda self: self._o.cons_len)
    msa_len = property(lambda self: self._o.msa_len)

    @property
    def cons_seq(self):
        if self._cons_seq is None:
            n = self._o.cons_len
            self._cons_seq = seqio.decode(np.frombuffer(C.string_at(self._o.cons_base, n), dtype=np.uint8), self._m) if n > 0 else ""
        return self._cons_seq

    @property
    def cons_cov(self):
        if self._cov_list is None:
            n = self._o.cons_len
            self._cov_list = np.frombuffer(C.string_at(self._o.cons_cov, 4 * n), dtype=np.int32).tolist() if n > 0 else []
        return self._cov_list

    @property
    def msa_seq(self):
        if self._msa_seq is None:
            o = self._o
            if o.msa_len > 0:
                rows = np.frombuffer(C.string_at(o.msa_base, o.msa_rows * o.msa_len), dtype=np.uint8).reshape(o.msa_rows, o.msa_len)
                self._msa_seq = [seqio.decode(row, self._m) for row in rows]
            else:
                self._msa_seq = []
        return self._msa_seq


class BatchResults(Sequence):
    """The results of one batch call as a read-only sequence of SetResult; the objects are made on first access (1000 of them cost
    more Python time than the call's own host work)."""
    __slots__ = ("_owner", "_m", "_items")

    def __init__(self, owner, m):
        self._owner, self._m, self._items = owner, m, [None] * owner.n

    def __len__(self):
        return len(self._items)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[j] for j in range(*i.indices(len(self._items)))]
        r = self._items[i]
        if r is None:
            if i < 0:
                i += len(self._items)
            r = self._items[i] = SetResult(self._owner, self._owner.out[i], self._m)
        return r


def _bind_msa(lib):
    if not getattr(lib, "_msa_bound", False):
        lib.abpoa_hip_msa_batch.argtypes = [C.POINTER(ffi.Scoring), C.c_int, C.POINTER(ReadSet), C.POINTER(Msa), C.c_uint, C.c_int]
        lib.abpoa_hip_msa_batch.restype = C.c_int
        lib.abpoa_hip_free_msa.argtypes = [C.POINTER(Msa)]
        if hasattr(lib, "abpoa_hip_free_msa_array"):
            lib.abpoa_hip_free_msa_array.argtypes = [C.POINTER(Msa), C.c_int]
            lib.abpoa_hip_free_msa_array.restype = None
        lib.abpoa_hip_get_msa_timing.argtypes = [C.POINTER(MsaTiming)]
        lib._msa_bound = True


class EncodedSets:
    """Read-sets encoded once into residue codes and laid out for abpoa_hip_msa_batch (kept alive on self)."""

    def __init__(self, read_sets, m=5, weights=None):
        """weights: None, or per set a list of per-read integer sequences (the reference's qv weights with -Q: seqio.qv_weights).
        A read given as a numpy uint8 array is taken as residue codes already (seqio.encode done elsewhere, e.g. in a generator process)."""
        self.n = len(read_sets)
        self.codes = [[np.ascontiguousarray(r if isinstance(r, np.ndarray) and r.dtype == np.uint8 else seqio.encode(r, m)) for r in rs] for rs in read_sets]
        self.sets = (ReadSet * self.n)()
        self._keep = []
        for i, rs in enumerate(self.codes):
            ptrs = (C.POINTER(C.c_uint8) * len(rs))(*[a.ctypes.data_as(C.POINTER(C.c_uint8)) for a in rs])
            lens = (C.c_int32 * len(rs))(*[len(a) for a in rs])
            wptrs = None
            if weights is not None and weights[i] is not None:
                ws = [np.ascontiguousarray(w, np.int32) for w in weights[i]]
                assert all(len(w) == len(a) for w, a in zip(ws, rs)), "one weight per base"
                wptrs = (C.POINTER(C.c_int32) * len(rs))(*[w.ctypes.data_as(C.POINTER(C.c_int32)) for w in ws])
                self._keep.append(ws)
            self._keep.append((ptrs, lens, wptrs))
            self.sets[i] = ReadSet(len(rs), ptrs, lens, wptrs)


def msa_batch(read_sets, params, out_cons=True, out_msa=False, n_threads=0, lib=None, encoded=None, weights=None, amb_strand=False):
    """Consensus / MSA of many independent read-sets (lists of strings) in one call.
    Returns a list of SetResult.  `lib` defaults to the HIP engine (tests may pass the CPU shim).
    weights: per set, per read, per base edge weights (the reference's -Q); amb_strand: the reference's -s."""
    lib = lib or ffi.lib()
    _bind_msa(lib)
    enc = encoded or EncodedSets(read_sets, params.m, weights)
    out = (Msa * enc.n)()
    sc = params.scoring()
    flags = (OUT_CONS if out_cons else 0) | (OUT_MSA if out_msa else 0) | (AMB_STRAND if amb_strand else 0)
    rc = lib.abpoa_hip_msa_batch(C.byref(sc), enc.n, enc.sets, out, flags, n_threads)
    if rc != 0:
        raise ffi.EngineError(f"abpoa_hip_msa_batch failed ({rc}): {lib.abpoa_hip_last_error().decode() if hasattr(lib, 'abpoa_hip_last_error') else ''}")
    res = BatchResults(_BatchOut(lib, out, enc.n), params.m)
    return res


class BatchContext:
    """abpoa_hip_ctx_t: the batch entry with per-caller state (own device queue, timing, last error): one host thread per context may call while others do."""

    def __init__(self, device=-1, lib=None):
        self.lib = lib or ffi.lib()
        _bind_msa(self.lib)
        self.lib.abpoa_hip_ctx_create.restype = C.c_void_p
        self.lib.abpoa_hip_ctx_create.argtypes = [C.c_int]
        self.lib.abpoa_hip_ctx_destroy.argtypes = [C.c_void_p]
        self.lib.abpoa_hip_msa_batch_ctx.argtypes = [C.c_void_p, C.POINTER(ffi.Scoring), C.c_int, C.POINTER(ReadSet), C.POINTER(Msa), C.c_uint, C.c_int]
        self.lib.abpoa_hip_msa_batch_ctx.restype = C.c_int
        self.lib.abpoa_hip_ctx_get_msa_timing.argtypes = [C.c_void_p, C.POINTER(MsaTiming)]
        self.lib.abpoa_hip_ctx_last_error.restype = C.c_char_p
        self.lib.abpoa_hip_ctx_last_error.argtypes = [C.c_void_p]
        self.h = self.lib.abpoa_hip_ctx_create(device)
        if not self.h:
            raise ffi.EngineError("abpoa_hip_ctx_create failed: " + self.lib.abpoa_hip_last_error().decode())

    def close(self):
        if self.h:
            self.lib.abpoa_hip_ctx_destroy(self.h)
            self.h = None

    __del__ = close

    def msa_batch(self, read_sets, params, out_cons=True, out_msa=False, n_threads=0, encoded=None, weights=None):
        enc = encoded or EncodedSets(read_sets, params.m, weights)
        out = (Msa * enc.n)()
        sc = params.scoring()
        flags = (OUT_CONS if out_cons else 0) | (OUT_MSA if out_msa else 0)
        rc = self.lib.abpoa_hip_msa_batch_ctx(self.h, C.byref(sc), enc.n, enc.sets, out, flags, n_threads)
        if rc != 0:
            raise ffi.EngineError(f"abpoa_hip_msa_batch_ctx failed ({rc}): {self.lib.abpoa_hip_ctx_last_error(self.h).decode()}")
        return BatchResults(_BatchOut(self.lib, out, enc.n), params.m)

    def timing(self):
        t = MsaTiming()
        self.lib.abpoa_hip_ctx_get_msa_timing(self.h, C.byref(t))
        return {k: getattr(t, k) for k, _ in MsaTiming._fields_}


def msa_timing(lib=None):
    lib = lib or ffi.lib()
    _bind_msa(lib)
    t = MsaTiming()
    lib.abpoa_hip_get_msa_timing(C.byref(t))
    d = {k: getattr(t, k) for k, _ in MsaTiming._fields_}
    d["pad"] = d["n_host_sets"]          # (the field's name before round 4)
    return d


HOST_REASONS = ["other", "node slots at the first read", "predecessor-list slots", "cigar slots", "node slots while fusing", "edge / aligned slots of a node",
                "projected graph growth", "row-order walk", "MSA rank walk", "DP arena too small for the bands", "other DP status", "job options / pass did not fit"]


def host_reasons(lib=None):
    """{reason: sets} for the read-sets the last msa_batch call handed to the host driver (abpoa_hip_get_host_reasons)."""
    lib = lib or ffi.lib()
    a = (C.c_int32 * 12)()
    lib.abpoa_hip_get_host_reasons.argtypes = [C.POINTER(C.c_int32)]
    lib.abpoa_hip_get_host_reasons.restype = None
    lib.abpoa_hip_get_host_reasons(a)
    return {HOST_REASONS[i]: int(a[i]) for i in range(12) if a[i]}


def format_output(result, names=None, out_cons=True, out_msa=False):
    """Text exactly as the reference prints it: abpoa_output_rc_msa (src/abpoa_output.c:70-101) when MSA is
    requested, else abpoa_output_fx_consensus (:495-512), single consensus."""
    lines = []
    if out_msa:
        if result.msa_len <= 0:
            return ""
        n_reads = len(result.msa_seq) - (1 if out_cons else 0)
        rc = result.is_rc
        for i in range(n_reads):
            nm = names[i] if names and i < len(names) and names[i] else None
            lines.append((f">{nm}" if nm else f">Seq_{i + 1}") + ("_reverse_complement" if rc[i] else ""))      # src/abpoa_output.c:77
            lines.append(result.msa_seq[i])
        if out_cons:
            lines.append(">Consensus_sequence")
            lines.append(result.msa_seq[-1])
    elif out_cons:
        lines.append(">Consensus_sequence")
        lines.append(result.cons_seq)
    return "\n".join(lines) + "\n"
