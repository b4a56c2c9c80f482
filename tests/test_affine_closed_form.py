"""CPU (numpy): the closed form of the F recurrence that every fast row loop uses (abpoa_amd/csrc/dp_common.h fast_f_chain, inj_dist; rows_fast.h's 64-lane prefix-max
form of the same) against a literal emulation of the reference's step (src/simd_abpoa_align.c:868-875 / :988-997 with SIMD_SET_F :665-682, set_num = pn):

    per vector:  F = ((H << 1) | first) - oe;   SIMD_SET_F(F);   first = max(H, F + o)[pn - 1]

where SIMD_SET_F is the log-step scan  F = max(F, ((F - 2^k e) << 2^k) | PRE_MIN[2^k])  that puts `inf` into the lanes it shifts in.

Claim: while no subtraction wraps,
    F[l] = max( max_{1 <= l' <= l} (H[l' - 1] - oe - (l - l') e) ,  first - oe - l e ,  inf - INJ[l] e )          INJ = dp_common.h inj_dist (none in the last lane)
    first' = max( H[pn - 1] , ownscan[pn - 1] + o , first - pn e )
-- the `inf` injections DO survive in F (unlike the linear-gap scan on H, tests/test_linear_closed_form.py: there `max(H, first)` clamps the lanes first), and the
table of their distances is what this test pins for both score widths."""
import numpy as np
import pytest

INJ = {16: [0] * 8 + [8] * 4 + [12] * 2 + [14, -1], 8: [0] * 4 + [4] * 2 + [6, -1]}      # dp_common.h inj_dist<16> / <8>
NEG = -(1 << 60)


def literal_vector(H, first, o, e, inf, pn):
    oe = o + e
    f = np.empty(pn, np.int64)
    f[0] = first - oe
    f[1:] = H[:pn - 1] - oe
    s = 1
    while s < pn:
        sh = np.full(pn, inf, np.int64)
        sh[s:] = f[:pn - s] - s * e
        f = np.maximum(f, sh)
        s *= 2
    return f, max(int(H[pn - 1]), int(f[pn - 1]) + o)


def closed_vector(H, first, o, e, inf, pn):
    oe = o + e
    own = np.full(pn, NEG, np.int64)
    for l in range(1, pn):
        own[l] = max(int(H[lp - 1]) - oe - (l - lp) * e for lp in range(1, l + 1))
    lane = np.arange(pn, dtype=np.int64)
    inj = np.array([inf - INJ[pn][l] * e if INJ[pn][l] >= 0 else NEG for l in range(pn)], np.int64)
    F = np.maximum(np.maximum(own, first - oe - lane * e), inj)
    return F, max(int(H[pn - 1]), int(own[pn - 1]) + o, first - pn * e)


@pytest.mark.parametrize("pn", [16, 8])
def test_f_closed_form_and_injection_table(pn):
    rng = np.random.default_rng(77 + pn)
    lo_t = -32768 if pn == 16 else -(1 << 31)
    for it in range(600):
        o = int(rng.integers(1, 25)); e = int(rng.integers(0, 5)); mis = int(rng.integers(1, 9))
        inf = lo_t + max(mis, o + e) + 31 * e                      # the reference's inf_min
        fast_lo = lo_t + (o + e) + pn * e                          # rows_fast.h fast_lo: below it the literal scan runs
        kind = rng.random(pn)
        H = int(rng.integers(-200, 800)) + rng.integers(-30, 31, pn).cumsum() // 2
        H = np.where(kind < 0.2, inf, H); H = np.where((kind >= 0.2) & (kind < 0.3), inf - mis, H)
        if it % 4 == 0:
            H[:int(rng.integers(1, pn + 1))] = inf                 # a dead stretch: the injections are all that F holds there
        H = np.maximum(H, fast_lo)
        first = int(H[0]) if rng.random() < 0.3 else int(rng.integers(inf, 900))
        lit, c1 = literal_vector(H.astype(np.int64), first, o, e, inf, pn)
        clo, c2 = closed_vector(H.astype(np.int64), first, o, e, inf, pn)
        assert np.array_equal(lit, clo), (pn, it, o, e, np.nonzero(lit != clo)[0][:4], lit[:pn], clo[:pn])
        assert c1 == c2
        assert lit.min() >= lo_t
