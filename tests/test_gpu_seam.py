"""GPU (-m gpu): drop-in proof for the narrowest seam.  oracle/_ref/abpoa_ref_gpu is the UNMODIFIED reference CLI /
graph / consensus code linked against libabpoa_hip.so instead of its own src/simd_abpoa_align.o (recipe:
oracle/Makefile); its output must be byte-identical to the pure reference binary on the same inputs, including the
paths that only the reference host code exercises (reverse-complement retry -s, seeding/sub-graph windows -S, MSA)."""
import os
import subprocess

import pytest

import helpers as H
from abpoa_amd import synth

pytestmark = pytest.mark.gpu
REF = os.path.join(H.REF_DIR, "abpoa_ref")
GPU = os.path.join(H.REF_DIR, "abpoa_ref_gpu")
D = os.path.join(H.GOLDEN_DIR, "data")


def _both(args):
    a = subprocess.run([REF] + args, capture_output=True, text=True, check=True).stdout
    b = subprocess.run([GPU] + args, capture_output=True, text=True)
    assert b.returncode == 0, b.stderr[-2000:]
    return a, b.stdout


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="prebuilt reference binaries not shipped")
@pytest.mark.parametrize("args", [
    ["-O", "4,0", "-E", "2", os.path.join(D, "seq.fa")],
    [os.path.join(D, "seq.fa")],
    ["-r", "2", os.path.join(D, "test.fa")],
    [os.path.join(D, "heter.fa")],
    ["-m", "1", "-r", "1", os.path.join(D, "heter.fa")],
    ["-m", "2", os.path.join(D, "heter.fa")],
    ["-O", "0,0", "-b", "-1", os.path.join(D, "seq.fa")],
    ["-s", os.path.join(D, "heter.fa")],
    ["-S", os.path.join(D, "heter.fa")],
    ["-r", "3", os.path.join(D, "test.fa")],
])
def test_stock_reference_with_gpu_dp_is_identical(args):
    a, b = _both(args)
    assert a == b and len(a) > 0


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="prebuilt reference binaries not shipped")
def test_seeded_windows_on_long_reads(tmp_path):
    """-S (minimizer seeding) turns every read into several sub-graph alignments: exercises beg/end windows,
    index_map and the band state carried between windows."""
    fa = str(tmp_path / "s.fa")
    synth.write_fasta(fa, synth.make_read_set(17, 0, 8, 3000, 0.05))
    a, b = _both(["-S", "-O", "4,0", "-E", "2", fa])
    assert a == b and len(a) > 3000


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(GPU)), reason="prebuilt reference binaries not shipped")
def test_seam_on_reads_with_ragged_ends(tmp_path):
    """The stock reference's own graph code over the GPU DP on reads cut at random places (local walks that reach predecessors hundreds of rows away, sources
    with dozens of successors, bands anchored off the path), every alignment mode and gap model, the strand retry and sub-graph windows: byte-identical to the
    pure reference binary."""
    import numpy as np
    rng = np.random.default_rng(211)
    comp = str.maketrans("ACGT", "TGCA")
    files = []
    for i, (n, ln, err) in enumerate([(30, 260, 0.04), (45, 420, 0.06), (20, 900, 0.05), (12, 2400, 0.08)]):
        reads = list(synth.make_read_set(23, i, n, ln, err))
        cut = [reads[0]]
        for r in reads[1:]:
            a = int(rng.integers(0, int(0.15 * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(0.15 * len(r)) + 1))
            cut.append(r[a:b])
        fa = str(tmp_path / f"r{i}.fa"); synth.write_fasta(fa, cut); files.append(fa)
        fr = str(tmp_path / f"rc{i}.fa"); synth.write_fasta(fr, [(r[::-1].translate(comp) if (j and j % 3 == 0) else r) for j, r in enumerate(cut)]); files.append(fr)
    for fa in files:
        rc_input = os.path.basename(fa).startswith("rc")
        for opts in ([["-s", "-r", "2"], ["-s", "-O", "4,0", "-E", "2"]] if rc_input else
                     [[], ["-r", "2"], ["-m", "1", "-r", "1"], ["-m", "1", "-O", "4,0", "-E", "2"], ["-m", "2", "-r", "2"], ["-O", "0,0", "-E", "2"], ["-b", "-1"], ["-S", "-r", "1"]]):
            a, b = _both(opts + [fa])
            assert a == b and len(a) > 0, (opts, fa)
