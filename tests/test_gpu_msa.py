"""GPU (-m gpu): the read-set batch API on the HIP engine must print byte-identical consensus / MSA text to the
reference (committed golden outputs; and, where the prebuilt reference binary travelled with the repo, the binary
itself run on the same seeded inputs), and survive ragged / degenerate batches."""
import os
import subprocess

import numpy as np
import pytest

import helpers as H
from abpoa_amd import api, ffi, seqio, synth

pytestmark = pytest.mark.gpu
D = H.GOLDEN_DIR
AG = dict(gap_open1=4, gap_open2=0, gap_ext1=2)


@pytest.fixture(scope="module", autouse=True)
def engine():
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1
    ffi.check(lib.abpoa_hip_init(0))
    return lib


def _run(fa, params, out_cons=True, out_msa=False):
    names, seqs = seqio.read_fasta(fa)
    r = api.msa_batch([seqs], params, out_cons=out_cons, out_msa=out_msa)[0]
    assert r.status == 0
    return api.format_output(r, names, out_cons, out_msa)


def _golden(name):
    return open(os.path.join(D, name, "output.txt")).read()


def test_config1_seq_fa():
    assert _run(os.path.join(D, "data", "seq.fa"), api.Params(**AG)) == _golden("out_seq_cons")


def test_readme_outputs():
    assert _run(os.path.join(D, "data", "test.fa"), api.Params(), False, True) == _golden("out_test_msa")
    assert _run(os.path.join(D, "data", "test.fa"), api.Params(), True, True) == _golden("out_test_cons_msa")
    assert _run(os.path.join(D, "data", "heter.fa"), api.Params()) == _golden("out_heter_cons")
    assert _run(os.path.join(D, "out_s1k_cons", "input.fa"), api.Params(**AG)) == _golden("out_s1k_cons")


def test_config5_shape_amino_acid_local_msa():
    p = api.Params(aln_mode=api.LOCAL, is_aa=True, score_matrix=os.path.join(D, "data", "BLOSUM62.mtx"))
    assert _run(os.path.join(D, "aa_blosum_loc", "input.fa"), p, False, True) == _golden("aa_blosum_loc")


def test_batch_of_sets_equals_oracle_backed_host_run():
    """64 read-sets (config-2 shape, shortened) in one lock-step batch: consensus + coverage + cell counts."""
    sets = [synth.make_read_set(2, i, 10, 400, 0.05) for i in range(64)]
    p = api.Params(**AG)
    got = api.msa_batch(sets, p, out_cons=True, out_msa=True)
    exp = api.msa_batch(sets, p, out_cons=True, out_msa=True, lib=H.cpu_shim_lib())
    for a, b in zip(got, exp):
        assert a.status == 0
        assert (a.cons_seq, a.cons_cov, a.msa_seq, a.n_cells) == (b.cons_seq, b.cons_cov, b.msa_seq, b.n_cells)


def test_ragged_and_degenerate_sets():
    sets = [synth.make_read_set(5, 0, 1, 50, 0.1), synth.make_read_set(5, 1, 2, 1, 0.0), synth.make_read_set(5, 2, 9, 333, 0.15),
            ["A", "C", "G", "T", "N"], ["ACGT" * 5] * 3]
    for pk in (AG, {}, dict(gap_open1=0, gap_open2=0)):
        p = api.Params(**pk)
        got = api.msa_batch(sets, p, out_cons=True, out_msa=True)
        exp = api.msa_batch(sets, p, out_cons=True, out_msa=True, lib=H.cpu_shim_lib())
        for a, b in zip(got, exp):
            assert (a.status, a.cons_seq, a.msa_seq) == (0, b.cons_seq, b.msa_seq)
    assert got[4].cons_seq == "ACGT" * 5


@pytest.mark.parametrize("cfg,opts,pk", [
    (2, ["-O", "4,0", "-E", "2"], AG),
    (3, [], {}),
])
def test_full_size_read_set_vs_reference_binary(tmp_path, cfg, opts, pk):
    """One full-size BASELINE read-set (config 2: 50 x 1 kb; config 3: 50 x 10 kb convex, int16 -> int32 switch)."""
    ref = os.path.join(H.REF_DIR, "abpoa_ref")
    if not os.path.exists(ref):
        pytest.skip("prebuilt reference binary not shipped")
    reads = synth.make_read_set(1, 0, **synth.CONFIGS[cfg])
    fa = str(tmp_path / "in.fa")
    synth.write_fasta(fa, reads)
    exp = subprocess.run([ref] + opts + [fa], capture_output=True, text=True, check=True).stdout
    r = api.msa_batch([reads], api.Params(**pk))[0]
    assert r.status == 0
    assert api.format_output(r) == exp


def _run_fx(path, params, out_cons, out_msa, qv=False, amb=False):
    names, seqs, quals = seqio.read_fastx(path)
    w = [[seqio.qv_weights(x, y) for x, y in zip(seqs, quals)]] if qv else None
    r = api.msa_batch([seqs], params, out_cons=out_cons, out_msa=out_msa, weights=w, amb_strand=amb)[0]
    assert r.status == 0
    return api.format_output(r, names, out_cons, out_msa), r


def test_ambiguous_strand_and_quality_weights():
    """reference -s (src/abpoa_align.c:315-336) and -Q (:462-467) through the engine: byte-identical text."""
    fa = os.path.join(D, "out_rc_cons", "input.fa")
    txt, r = _run_fx(fa, api.Params(**AG), True, False, amb=True)
    assert txt == _golden("out_rc_cons") and [i for i, f in enumerate(r.is_rc) if f] == [2, 5, 8]
    assert _run_fx(fa, api.Params(), True, True, amb=True)[0] == _golden("out_rc_msa")
    txt, r = _run_fx(os.path.join(D, "out_rc_long_msa", "input.fa"), api.Params(), True, True, amb=True)      # long noisy reads: the retry inherits the forward pass's band state
    assert txt == _golden("out_rc_long_msa") and [i for i, f in enumerate(r.is_rc) if f] == [2, 5]
    fq = os.path.join(D, "out_qv_cons", "input.fq")
    assert _run_fx(fq, api.Params(**AG), True, False, qv=True)[0] == _golden("out_qv_cons")
    assert _run_fx(fq, api.Params(), True, True, qv=True)[0] == _golden("out_qv_msa")


@pytest.mark.parametrize("wl,idx", [("cfg2", [0, 1, 2, 3, 999]), ("cfg4", [0, 1]), ("cfg3", [0, 1]), ("cfg5", [0, 1, 2, 3, 999])])
def test_full_size_sets_against_committed_reference_digests(wl, idx):
    """Full-size BASELINE read-sets (50 x 1 kb; 50 x 10 kb affine 5 % and convex 15 %; 30 x 500 aa local BLOSUM62 MSA): the output text of
    each set must hash to what the reference printed for it (tests/golden/bench_digests/, oracle/make_bench_digests.py)."""
    from abpoa_amd import workloads as W
    w = W.WORKLOADS[wl]
    ref = W.load_digests(wl)
    sets = [synth.make_read_set(1, i, **synth.CONFIGS[w["cfg"]]) for i in idx]
    res = api.msa_batch(sets, api.Params(**w["params"]), out_cons=not w["out_msa"], out_msa=w["out_msa"])
    for i, r in zip(idx, res):
        assert r.status == 0
        txt = api.format_output(r, [f"r{j}" for j in range(len(sets[0]))], not w["out_msa"], w["out_msa"])
        assert W.output_sha(txt) == ref[i], f"{wl} set {i}"
