"""CPU: the direction plane (abpoa_amd/csrc/dir_plane.h) is a complete record of the reference backtrack's decisions.

oracle/dir_model.c builds, from a full oracle trace, the per-cell words exactly as the HIP row loops define them and walks them in the
reference's order; the cigar and every abpoa_res_t field must equal those of the oracle's value-comparing backtrack (itself pinned against
the compiled reference) -- on every golden alignment the plane applies to, and on every alignment of seeded read-sets (the model rides
inside the oracle-backed host driver with ABPOA_SHIM_DIR_CHECK=1).  The model also counts where the cheap rules the row loops use (F
origin derived from the left neighbour's H - F, uE from the E update's own maximum) disagree with the literal comparisons: never, on
cells that hold real scores."""
import os

import numpy as np
import pytest

import helpers as H
from abpoa_amd import api, synth


def _check(label, case):
    rc, a, b, st = H.run_dir_model(case)
    if rc != 0:
        return None
    for k in a:
        if k == "cigar":
            assert np.array_equal(a[k], b[k]), f"{label}: cigar differs (first at word {int(np.nonzero(a[k][:len(b[k])] != b[k][:len(a[k])])[0][0]) if len(a[k]) == len(b[k]) else -1}, lengths {len(a[k])} / {len(b[k])})"
        else:
            assert a[k] == b[k], f"{label}: {k} differs: oracle {a[k]} model {b[k]}"
    assert st[2] == 0, f"{label}: derived F origin differs from the literal comparison on {st[2]} cells with real scores"
    assert st[3] == 0, f"{label}: arithmetic uE differs from the literal comparison on {st[3]} cells with real scores"
    assert st[7] == 0, f"{label}: {st[7]} walk steps took an F origin the reference's comparisons do not give"
    return st


def test_goldens_walk_the_plane():
    n_applied, tot = 0, np.zeros(10, np.int64)
    for label, path in H.golden_cases():
        g = H.read_abpg(path)
        st = _check(label, H.FlatCase(g))
        if st is not None:
            n_applied += 1
            tot += np.array(st)
    assert n_applied >= 20, n_applied           # global, banded, affine / convex goldens incl. the 10 kb / 20 kb int32 ones
    assert tot[4] > 40000 and tot[1] > 0        # steps walked; cells in masked-scan vectors exist (their literal override is exercised)


@pytest.mark.parametrize("name,params,shape", [
    ("affine 1 kb 5 %", dict(gap_open1=4, gap_open2=0, gap_ext1=2), (12, 1000, 0.05)),
    ("convex 1 kb 15 %", dict(), (12, 1000, 0.15)),
    ("convex 3 kb 15 %", dict(), (6, 3000, 0.15)),
    ("affine o=7 e=1", dict(gap_open1=7, gap_open2=0, gap_ext1=1), (10, 600, 0.1)),
    ("convex o=1,31 e=3,1", dict(gap_open1=1, gap_ext1=3, gap_open2=31, gap_ext2=1), (10, 600, 0.1)),
])
def test_every_alignment_of_seeded_read_sets(name, params, shape, monkeypatch):
    """the model runs beside every oracle alignment of the oracle-backed host driver and aborts the run on the first difference"""
    monkeypatch.setenv("ABPOA_SHIM_DIR_CHECK", "1")
    sets = [synth.make_read_set(41, i, *shape) for i in range(3)]
    res = api.msa_batch(sets, api.Params(**params), lib=H.cpu_shim_lib(), n_threads=2)
    assert all(r.status == 0 for r in res), name
    n = H.cpu_shim_lib().abpoa_shim_dir_checked()
    assert n >= 3 * (shape[0] - 1), (name, n)
