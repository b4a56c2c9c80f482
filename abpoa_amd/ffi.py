"""ctypes view of the C-ABI in include/abpoa_hip.h and loader of the in-tree engine library.

The library is the product: if abpoa_amd/libabpoa_hip.so is missing this module raises at load time
(there is no Python or CPU fallback for the DP)."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ABPOA_HIP_LIB") or os.path.join(_HERE, "libabpoa_hip.so")   # env override: profiling builds (tools/)


class Scoring(C.Structure):          # abpoa_hip_scoring_t
    _fields_ = [("m", C.c_int32), ("mat", C.POINTER(C.c_int32)), ("max_mat", C.c_int32), ("min_mis", C.c_int32),
                ("gap_open1", C.c_int32), ("gap_ext1", C.c_int32), ("gap_open2", C.c_int32), ("gap_ext2", C.c_int32),
                ("align_mode", C.c_int32), ("gap_mode", C.c_int32), ("wb", C.c_int32), ("wf", C.c_float),
                ("zdrop", C.c_int32), ("ret_cigar", C.c_int32), ("rev_cigar", C.c_int32)]


class Problem(C.Structure):          # abpoa_hip_problem_t
    _fields_ = [("n_rows", C.c_int32), ("qlen", C.c_int32), ("query", C.POINTER(C.c_uint8)),
                ("row_base", C.POINTER(C.c_uint8)), ("row_node_id", C.POINTER(C.c_int32)),
                ("row_remain", C.POINTER(C.c_int32)), ("row_active", C.POINTER(C.c_uint8)),
                ("pred_off", C.POINTER(C.c_int32)), ("pred_row", C.POINTER(C.c_int32)),
                ("out_off", C.POINTER(C.c_int32)), ("out_row", C.POINTER(C.c_int32)),
                ("max_pos_left", C.POINTER(C.c_int32)), ("max_pos_right", C.POINTER(C.c_int32))]


class Trace(C.Structure):            # abpoa_hip_trace_t
    _fields_ = [("bits", C.c_int32), ("n_planes", C.c_int32),
                ("dp_beg", C.POINTER(C.c_int32)), ("dp_end", C.POINTER(C.c_int32)),
                ("dp_beg_sn", C.POINTER(C.c_int32)), ("dp_end_sn", C.POINTER(C.c_int32)),
                ("row_off", C.POINTER(C.c_int64)), ("planes", C.c_void_p), ("row_max_i", C.POINTER(C.c_int32))]


class Result(C.Structure):           # abpoa_hip_result_t
    _fields_ = [("status", C.c_int32), ("bits", C.c_int32), ("best_score", C.c_int32), ("best_row", C.c_int32),
                ("best_col", C.c_int32), ("node_s", C.c_int32), ("node_e", C.c_int32), ("query_s", C.c_int32),
                ("query_e", C.c_int32), ("n_aln_bases", C.c_int32), ("n_matched_bases", C.c_int32),
                ("n_cigar", C.c_int32), ("cigar", C.POINTER(C.c_uint64)), ("n_cells", C.c_int64),
                ("trace", C.POINTER(Trace))]


class Stats(C.Structure):            # abpoa_hip_stats_t
    _fields_ = [("n_launches", C.c_int64), ("n_alignments", C.c_int64), ("n_cells", C.c_int64),
                ("algo_bytes", C.c_int64), ("kernel_ms", C.c_double), ("h2d_ms", C.c_double), ("d2h_ms", C.c_double), ("tail_ms", C.c_double),
                ("rounds_ms", C.c_double), ("rounds_launches", C.c_int64), ("rounds_algo_bytes", C.c_int64)]


FLAG_TRACE = 0x1
# every symbol include/abpoa_hip.h declares (checked by tests/test_abi.py without a GPU)
EXPORTS = ["abpoa_hip_device_count", "abpoa_hip_init", "abpoa_hip_shutdown", "abpoa_hip_last_error",
           "abpoa_hip_get_stats", "abpoa_hip_reset_stats", "abpoa_hip_align_batch", "abpoa_hip_free_result",
           "abpoa_hip_score_bits", "abpoa_hip_trim"]

_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(the HIP engine is the only compute path; there is no fallback)")
        L = C.CDLL(LIB_PATH)
        L.abpoa_hip_device_count.restype = C.c_int
        L.abpoa_hip_init.argtypes = [C.c_int]
        L.abpoa_hip_init.restype = C.c_int
        L.abpoa_hip_last_error.restype = C.c_char_p
        L.abpoa_hip_get_stats.argtypes = [C.POINTER(Stats)]
        L.abpoa_hip_align_batch.argtypes = [C.POINTER(Scoring), C.c_int, C.POINTER(Problem), C.POINTER(Result), C.c_uint]
        L.abpoa_hip_align_batch.restype = C.c_int
        L.abpoa_hip_free_result.argtypes = [C.POINTER(Result)]
        L.abpoa_hip_score_bits.argtypes = [C.POINTER(Scoring), C.c_int, C.c_int, C.POINTER(C.c_int32)]
        L.abpoa_hip_score_bits.restype = C.c_int
        _lib = L
    return _lib


class EngineError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise EngineError(f"abpoa_hip error {rc}: {lib().abpoa_hip_last_error().decode()}")


def stats():
    s = Stats()
    lib().abpoa_hip_get_stats(C.byref(s))
    return {k: getattr(s, k) for k, _ in Stats._fields_}
