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


def test_wide_loop_lds_plan_respects_the_allocation_granule():
    """Host logic, no GPU: the LDS plan of the wide row loop.  gfx950 hands LDS out in pieces of 1280 bytes, 128 to a CU (tools/probes/lds_granule.hip), so
    n workgroups share a CU only if each needs at most floor(128 / n) pieces; the plan takes the deepest score ring with which the whole launch is
    resident, counting at most eight workgroups per CU, and the wide kernels' own carve-up packs the query two codes to a byte."""
    lib = ffi.lib()
    out = (C.c_int * 8)()
    for kw in (dict(), dict(gap_open1=4, gap_open2=0, gap_ext1=2)):
        sc = api.Params(**kw).scoring()
        seen = {}
        for bits in (16, 32):
            for n_aln in (64, 256, 512, 768, 1024, 1100, 2048, 5000):
                lib.abpoa_hip__wide_plan(C.byref(sc), 10100, bits, n_aln, out)
                nw, rows, total, per_cu, wph, lo, hi = out[0], out[1], out[2], out[3], out[4], out[5], out[6]
                assert nw == 1 and rows in (4, 8, 16) and lo <= 10 + 101 <= hi
                assert per_cu >= 1 and ((total + 1279) // 1280) * per_cu <= 128, (kw, bits, n_aln, list(out))
                assert per_cu * 256 >= min(n_aln, 2048) or rows == 4, (kw, bits, n_aln, list(out))      # resident, or the ring is as shallow as it gets
                assert wph <= (10100 + 2) // 2 + 16 + 4 * sc.m * (sc.m + 1) + 32                               # 4-bit query codes
                seen[(bits, n_aln)] = (rows, per_cu)
        assert seen[(32, 1024)] == (8, 4) and seen[(32, 2048)] == (4, 8), seen      # 10 kb reads: four per CU with the 8-row ring, eight with the 4-row ring
        assert seen[(32, 64)][0] == 16


def test_no_device_fails_loudly():
    lib = ffi.lib()
    if lib.abpoa_hip_device_count() > 0:
        pytest.skip("a GPU is present")
    assert lib.abpoa_hip_init(0) == -1          # ABPOA_HIP_ENODEV
    assert b"no HIP device" in lib.abpoa_hip_last_error()
    with pytest.raises(ffi.EngineError):
        api.msa_batch([["ACGT", "ACGT"]], api.Params())


def test_switches_form_one_table_and_nothing_else_reads_the_environment():
    """Round 5 hygiene: every behaviour switch of the library is a row of abpoa_amd/csrc/engine_options.cpp, read once per C-ABI entry into
    a snapshot (environment, overridden by abpoa_hip_set_option); no other source file of the engine calls getenv."""
    import glob
    lib = ffi.lib()
    lib.abpoa_hip_list_options.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_int]; lib.abpoa_hip_list_options.restype = C.c_int
    n = lib.abpoa_hip_list_options(None, None, 0)
    names, helps = (C.c_char_p * n)(), (C.c_char_p * n)()
    assert lib.abpoa_hip_list_options(names, helps, n) == n and n >= 30
    table = {names[i].decode(): helps[i].decode() for i in range(n)}
    assert "ABPOA_HIP_STRICT" in table and "ABPOA_GPU_DEVICES" in table and all(h[:1] in "PTD" for h in table.values())
    lib.abpoa_hip_set_option.argtypes = [C.c_char_p, C.c_char_p]
    assert lib.abpoa_hip_set_option(b"ABPOA_HIP_NO_SUCH_SWITCH", b"1") != 0
    assert lib.abpoa_hip_set_option(b"ABPOA_HIP_VERBOSE", b"1") == 0 and lib.abpoa_hip_set_option(b"ABPOA_HIP_VERBOSE", None) == 0
    csrc = os.path.join(H.ROOT, "abpoa_amd", "csrc")
    used = set()
    for fn in glob.glob(os.path.join(csrc, "*.cpp")) + glob.glob(os.path.join(csrc, "*.hip")) + glob.glob(os.path.join(csrc, "*.h")):
        text = open(fn).read()
        if not fn.endswith("engine_options.cpp"):
            assert "getenv(" not in text.replace("opt_env(", ""), f"{os.path.basename(fn)} reads the environment directly"
            used |= set(re.findall(r'(?:opt_env|env_int|env_on)\("(ABPOA_[A-Z_0-9]+)"', text))
    assert used <= set(table), f"switches read but not in the table: {sorted(used - set(table))}"
