"""CPU (numpy): the closed form behind the linear-gap row of the fast loops (abpoa_amd/csrc/rows_fast.h GAP = 0, rows_general.h linear_h) against a literal
emulation of the reference's in-row step (src/simd_abpoa_align.c:762-778): per SIMD vector `H = max(H, first)`, `SIMD_SET_F(H, ...)` with set_num = pn
(:665-682: log-step shifts of H - 2^k e with `inf` in the lanes shifted in), `first = {H[pn - 1] - e, inf, ...}`.

Claim: while no subtraction wraps, over a whole 64-lane chunk
    H[c] = max( max_{c' <= c} (h[c'] + c' e) - c e ,  inf )
with the carry entering at lane 0.  The clamp at `inf` is the reference's own: `first` holds `inf` in its lanes 1 .. pn - 1, so `max(H, first)` lifts every
such lane to `inf` before the scan, and lane 0 gets `inf` from the scan's first shifted-in lane -- unlike the affine F scan, whose lanes keep
`inf - INJ e` (dp_common.h inj_dist).  The first version of the closed form used that affine term and was wrong by up to pn e in stretches of a row that no
real score reaches (lanes pn / 2 .. pn - 2 of a vector whose left neighbours are all `inf`-like) -- cells no path uses, found by this test, not by the
goldens.  Checked on random rows that mix real scores, `inf`, values slightly below `inf` (`inf + q` of a mismatch, `inf - e` of a vertical step) and dead
stretches of such values, for both score widths."""
import numpy as np
import pytest

def literal_chunk(h, first, e, inf, pn):
    """The reference's loop over the 64 / pn vectors of a chunk: returns (H, carry out)."""
    out = np.empty(64, np.int64)
    for v in range(64 // pn):
        f = h[v * pn:(v + 1) * pn].astype(np.int64).copy()
        f[0] = max(f[0], first)                       # dp_h = max(dp_h, first): first = {carry, inf, inf, ...} and every score is >= ... lanes 1.. see below
        f[1:] = np.maximum(f[1:], inf)                # (the other lanes of `first` hold inf)
        s = 1
        while s < pn:                                 # SIMD_SET_F, set_num == pn
            sh = np.full(pn, inf, np.int64)           # zero-filled shift | PRE_MIN: `inf` in the s lanes shifted in
            sh[s:] = f[:pn - s] - s * e
            f = np.maximum(f, sh)
            s *= 2
        out[v * pn:(v + 1) * pn] = f
        first = f[pn - 1] - e
    return out, first


def closed_chunk(h, first, e, inf, pn):
    lane = np.arange(64, dtype=np.int64)
    g = h.astype(np.int64) + lane * e
    g[0] = max(g[0], first)
    pre = np.maximum.accumulate(g) - lane * e
    H = np.maximum(pre, inf)
    return H, H[63] - e


@pytest.mark.parametrize("pn", [16, 8])
def test_one_prefix_max_scan_equals_the_vector_by_vector_scan(pn):
    rng = np.random.default_rng(20261005 + pn)
    lo_t = -32768 if pn == 16 else -(1 << 31)
    for it in range(400):
        e = int(rng.integers(1, 6)); mis = int(rng.integers(1, 9))
        inf = lo_t + max(mis, e) + 31 * e             # the reference's inf_min (src/simd_abpoa_align.c:1676-1680) for a linear job
        fast_lo = lo_t + e + pn * e                   # the closed form's guard (rows_fast.h fast_lo): below it the literal scan runs
        kind = rng.random(64)
        base = int(rng.integers(-300, 900))
        h = base + rng.integers(-40, 41, 64).cumsum() // 3
        h = np.where(kind < 0.15, inf, h)             # cells no predecessor reaches
        h = np.where((kind >= 0.15) & (kind < 0.25), inf - mis, h)      # inf + q of a mismatch
        h = np.where((kind >= 0.25) & (kind < 0.30), inf - e, h)        # inf - e of a vertical step
        if it % 3 == 0:                               # a dead stretch at the left of the chunk (no real score reaches it): only inf-like values for 5 .. 40 lanes
            nd = int(rng.integers(5, 41)); dead = rng.random(nd)
            h[:nd] = np.where(dead < 0.4, inf, np.where(dead < 0.7, inf - mis, inf - e))
        assert h.min() >= fast_lo
        first = int(h[0]) if rng.random() < 0.5 else (inf - e if it % 3 == 0 else int(rng.integers(inf - e, base + 50)))      # chunk 0: the row's own first column; later chunks: a carry
        lit, c1 = literal_chunk(h, first, e, inf, pn)
        clo, c2 = closed_chunk(h, first, e, inf, pn)
        assert np.array_equal(lit, clo), (pn, it, e, np.nonzero(lit != clo)[0][:4])
        assert c1 == c2
        assert lit.min() >= lo_t                       # nothing wrapped on the way: the literal values are the int16 / int32 values
