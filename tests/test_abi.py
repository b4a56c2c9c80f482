"""CPU: the C-ABI shared library loads without a GPU and exports every symbol include/abpoa_hip.h declares;
without a device the compute entry points fail loudly (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import helpers as H
from abpoa_amd import api, ffi


def _declared_symbols():
    src = open(os.path.join(H.ROOT, "include", "abpoa_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(abpoa_hip_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = ffi.lib()
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/abpoa_hip.h but not exported"
    for s in ffi.EXPORTS:
        assert s in syms


def test_struct_sizes_match_header(tmp_path):
    """ctypes mirrors vs the C compiler's view of include/abpoa_hip.h (sizes and a few offsets)."""
    import subprocess
    src = tmp_path / "sz.c"
    src.write_text("""#include <stdio.h>
#include <stddef.h>
#include "abpoa_hip.h"
int main(void) { printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(abpoa_hip_scoring_t), sizeof(abpoa_hip_problem_t),
  sizeof(abpoa_hip_result_t), sizeof(abpoa_hip_trace_t), sizeof(abpoa_hip_stats_t), sizeof(abpoa_hip_readset_t), sizeof(abpoa_hip_msa_t),
  sizeof(abpoa_hip_msa_timing_t), offsetof(abpoa_hip_result_t, cigar), offsetof(abpoa_hip_msa_t, msa_base)); return 0; }""")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I" + os.path.join(H.ROOT, "include"), "-o", str(exe), str(src)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    exp = [C.sizeof(ffi.Scoring), C.sizeof(ffi.Problem), C.sizeof(ffi.Result), C.sizeof(ffi.Trace), C.sizeof(ffi.Stats),
           C.sizeof(api.ReadSet), C.sizeof(api.Msa), C.sizeof(api.MsaTiming), ffi.Result.cigar.offset, api.Msa.msa_base.offset]
    assert got == exp


def test_score_bits_matches_oracle_without_gpu():
    lib, olib = ffi.lib(), H.oracle_lib()
    p = api.Params()
    sc = p.scoring()
    inf_a, inf_b = C.c_int32(), C.c_int32()
    for gn, ql in ((10, 10), (16364, 100), (16365, 100), (100, 20000), (3, 0)):
        assert lib.abpoa_hip_score_bits(C.byref(sc), gn, ql, C.byref(inf_a)) == olib.abpoa_oracle_score_bits(C.byref(sc), gn, ql, C.byref(inf_b))
        assert inf_a.value == inf_b.value


def test_no_device_fails_loudly():
    lib = ffi.lib()
    if lib.abpoa_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    assert lib.abpoa_hip_init(0) == -1          # ABPOA_HIP_ENODEV
    assert b"no HIP device" in lib.abpoa_hip_last_error()
    with pytest.raises(ffi.EngineError):
        api.msa_batch([["ACGT", "ACGT"]], api.Params())
