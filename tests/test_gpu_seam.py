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
