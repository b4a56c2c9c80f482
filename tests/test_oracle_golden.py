"""CPU: the plain-C oracle (oracle/abpoa_dp_oracle.c) must reproduce every committed golden vector
(bands, score planes or their per-row checksums, best score, cigar, band state) bit for bit.
The goldens were generated from the compiled reference by oracle/make_golden.py."""
import pytest

import helpers as H

CASES = H.golden_cases()


def test_golden_fixtures_present():
    assert len(CASES) >= 60


@pytest.mark.parametrize("label,path", CASES, ids=[c[0] for c in CASES])
def test_oracle_matches_golden(label, path):
    g = H.read_abpg(path)
    case = H.FlatCase(g)
    o = H.run_oracle(case)
    H.compare_with_golden(o, g, label=label)


def test_score_bits_switch():
    """int16 -> int32 switch point, reference src/simd_abpoa_align.c:1672-1683 (defaults: convex 4,24/2,1, mismatch 4)."""
    import ctypes as C
    import numpy as np
    lib = H.oracle_lib()
    mat = np.zeros(25, np.int32)
    sc = H.Scoring(5, mat.ctypes.data_as(C.POINTER(C.c_int32)), 2, 4, 4, 2, 24, 1, 0, 2, 10, 0.01, -1, 1, 0)
    inf = C.c_int32()
    # max_score = max(qlen*2, max(qlen,gn)*2+4) <= 32767-4-6-25 = 32732  <=> len <= 16364
    assert lib.abpoa_oracle_score_bits(C.byref(sc), 16364, 100, C.byref(inf)) == 16
    assert inf.value == -32768 + 25 + 31 * 2
    assert lib.abpoa_oracle_score_bits(C.byref(sc), 16365, 100, C.byref(inf)) == 32
    assert inf.value == -2147483648 + 25 + 62
