"""CPU, development container only: diff the oracle against the compiled reference (oracle/_ref/ref_dump,
built from /root/reference by oracle/Makefile) on a matrix of gap modes x align modes x band on/off x
score width, cell for cell.  Skipped where the reference build is absent and nothing here runs on the
GPU box (no /root/reference there)."""
import glob
import os

import pytest

import helpers as H
from abpoa_amd import synth

pytestmark = pytest.mark.skipif(not (H.have_ref() and os.path.isdir(H.REFERENCE_TREE)), reason="reference build not available")

AG, CG, LG = ["-O", "4,0", "-E", "2"], [], ["-O", "0,0", "-E", "2"]
MODES = {"gb": [], "gu": ["-b", "-1"], "loc": ["-m", "1"], "ext": ["-m", "2"], "extz": ["-m", "2", "-z", "5"]}


def _check(tmp_path, name, fa, opts, reads, sub=None):
    d = str(tmp_path / name)
    H.run_ref_dump(fa, d, opts, reads, planes=1, sub=sub)
    files = sorted(glob.glob(d + "/*.abpg"))
    assert files
    for f in files:
        g = H.read_abpg(f)
        o = H.run_oracle(H.FlatCase(g))
        H.compare_with_golden(o, g, label=name + "/" + os.path.basename(f))


@pytest.mark.parametrize("gap", ["ag", "cg", "lg"])
@pytest.mark.parametrize("mode", list(MODES))
def test_reference_test_data(tmp_path, gap, mode):
    g = {"ag": AG, "cg": CG, "lg": LG}[gap]
    _check(tmp_path, "seq", os.path.join(H.REFERENCE_TREE, "test_data/seq.fa"), g + MODES[mode], "all")
    _check(tmp_path, "het", os.path.join(H.REFERENCE_TREE, "test_data/heter.fa"), g + MODES[mode], "2,7,14")


@pytest.mark.parametrize("gap", ["ag", "cg", "lg"])
def test_synthetic_1kb(tmp_path, gap):
    fa = str(tmp_path / "s.fa")
    synth.write_fasta(fa, synth.make_read_set(7, 3, 10, 1000, 0.05))
    g = {"ag": AG, "cg": CG, "lg": LG}[gap]
    _check(tmp_path, "b", fa, g, "1,9")
    _check(tmp_path, "u", fa, g + ["-b", "-1"], "4")


@pytest.mark.parametrize("e1,wb", [(2, 10), (1, 3), (4, 2), (3, 6)], ids=["e2_b10", "e1_b3", "e4_b2", "e3_b6"])
def test_linear_gaps_with_narrow_bands(tmp_path, e1, wb):
    """Linear gaps with bands of -b 2 .. 10 -f 0: rows with stretches no real score reaches (the reference clamps those lanes at `inf`, src/simd_abpoa_align.c:773-775) --
    the regime of tests/test_gpu_parity.py::test_linear_rows_plane_level_on_the_golden_graphs, which compares the fast row loops with this oracle."""
    g = ["-O", "0,0", "-E", str(e1), "-b", str(wb), "-f", "0"]
    _check(tmp_path, "seq", os.path.join(H.REFERENCE_TREE, "test_data/seq.fa"), g, "all")
    _check(tmp_path, "het", os.path.join(H.REFERENCE_TREE, "test_data/heter.fa"), g, "2,7,14")
    fa = str(tmp_path / "s.fa")
    synth.write_fasta(fa, synth.make_read_set(17, 2, 8, 700, 0.12))
    _check(tmp_path, "syn", fa, g, "3,7")


def test_synthetic_10kb_int32(tmp_path):
    fa = str(tmp_path / "s.fa")
    synth.write_fasta(fa, synth.make_read_set(11, 0, 9, 10000, 0.15))
    _check(tmp_path, "cg", fa, CG, "8")
    _check(tmp_path, "ag", fa, AG, "8")


def test_amino_acid_blosum62(tmp_path):
    fa = str(tmp_path / "aa.fa")
    synth.write_fasta(fa, synth.make_read_set(13, 1, 8, 500, alphabet=synth.AA, rates=(0.05, 0.03, 0.03)))
    mtx = os.path.join(H.REFERENCE_TREE, "BLOSUM62.mtx")
    _check(tmp_path, "loc", fa, ["-m", "1", "-c", "-t", mtx], "2,7")
    _check(tmp_path, "glob", fa, ["-c", "-t", mtx], "5")


def test_subgraph(tmp_path):
    _check(tmp_path, "s1", os.path.join(H.REFERENCE_TREE, "test_data/seq.fa"), AG, "3,5,8", sub=(10, 40))
    _check(tmp_path, "s2", os.path.join(H.REFERENCE_TREE, "test_data/heter.fa"), CG, "4,9", sub=(100, 500))
