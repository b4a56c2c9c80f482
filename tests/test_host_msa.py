"""CPU: the product's HOST layer (graph fusion, row order, consensus, RC-MSA, read-set driver; built into
tests/_build/libcpu_shim.so with an oracle-backed aligner) must print exactly what the reference prints.
Expected texts are the reference's outputs committed under tests/golden/out_*/output.txt."""
import os
import subprocess

import pytest

import helpers as H
from abpoa_amd import api, seqio, synth

D = H.GOLDEN_DIR
AG = dict(gap_open1=4, gap_open2=0, gap_ext1=2)


def _run(fa, params, out_cons=True, out_msa=False, n_threads=2):
    names, seqs = seqio.read_fasta(fa)
    r = api.msa_batch([seqs], params, out_cons=out_cons, out_msa=out_msa, lib=H.cpu_shim_lib(), n_threads=n_threads)[0]
    assert r.status == 0
    return api.format_output(r, names, out_cons, out_msa), r


def _golden(name):
    return open(os.path.join(D, name, "output.txt")).read()


def test_seq_fa_affine_consensus():           # BASELINE.json config 1
    txt, r = _run(os.path.join(D, "data", "seq.fa"), api.Params(**AG))
    assert txt == _golden("out_seq_cons")
    assert r.cons_seq == "CGTCAATCTATCGAAGCATACGCGGCAGAGCCGAAGACCTCGGCAATCAC"
    assert r.cons_cov[:10] == [10, 10, 10, 10, 10, 10, 10, 9, 9, 10]     # pyabpoa golden (SURVEY.md 8c iii)


def test_readme_msa_and_consensus():           # reference README.md:169-195
    txt, _ = _run(os.path.join(D, "data", "test.fa"), api.Params(), out_cons=False, out_msa=True)
    assert txt == _golden("out_test_msa")
    assert txt.split("\n")[1::2][:4] == ["ACGTGTACA-GTTGAC", "A-G-GTACACGTT-AC", "A-GTGT-CACGTTGAC", "ACGTGTACA--TTGAC"]
    txt, _ = _run(os.path.join(D, "data", "test.fa"), api.Params(), out_cons=True, out_msa=True)
    assert txt == _golden("out_test_cons_msa")


def test_heter_convex_consensus():
    txt, _ = _run(os.path.join(D, "data", "heter.fa"), api.Params())
    assert txt == _golden("out_heter_cons")


def test_synthetic_1kb_consensus():
    txt, _ = _run(os.path.join(D, "out_s1k_cons", "input.fa"), api.Params(**AG))
    assert txt == _golden("out_s1k_cons")


def test_amino_acid_blosum62_local_msa():       # BASELINE.json config 5 shape (small)
    p = api.Params(aln_mode=api.LOCAL, is_aa=True, score_matrix=os.path.join(D, "data", "BLOSUM62.mtx"))
    txt, _ = _run(os.path.join(D, "aa_blosum_loc", "input.fa"), p, out_cons=False, out_msa=True)
    assert txt == _golden("aa_blosum_loc")


def _run_fx(path, params, out_cons, out_msa, qv=False, amb=False, lib=None):
    names, seqs, quals = seqio.read_fastx(path)
    w = [[seqio.qv_weights(x, y) for x, y in zip(seqs, quals)]] if qv else None
    r = api.msa_batch([seqs], params, out_cons=out_cons, out_msa=out_msa, lib=lib or H.cpu_shim_lib(), n_threads=2, weights=w, amb_strand=amb)[0]
    assert r.status == 0
    return api.format_output(r, names, out_cons, out_msa), r


def test_ambiguous_strand_retry():              # reference -s, src/abpoa_align.c:315-336
    fa = os.path.join(D, "out_rc_cons", "input.fa")
    txt, r = _run_fx(fa, api.Params(**AG), True, False, amb=True)
    assert txt == _golden("out_rc_cons")
    assert [i for i, f in enumerate(r.is_rc) if f] == [2, 5, 8]
    txt, _ = _run_fx(fa, api.Params(), True, True, amb=True)
    assert txt == _golden("out_rc_msa")             # (names carry _reverse_complement, src/abpoa_output.c:77)
    no_retry, _ = _run_fx(fa, api.Params(**AG), True, False)
    assert no_retry != _golden("out_rc_cons")       # the retry matters on this input


def test_ambiguous_strand_retry_long_noisy_reads():
    """8 x 2.5 kb reads at 15 % error, two of them reverse complements, default adaptive band: the retry DP starts from the band bounds the forward
    DP pushed (the reference re-sorts -- and resets them -- only once per read, src/abpoa_align.c:329 / abpoa_graph.c:303-308)."""
    fa = os.path.join(D, "out_rc_long_msa", "input.fa")
    txt, r = _run_fx(fa, api.Params(), True, True, amb=True)
    assert txt == _golden("out_rc_long_msa") and [i for i, f in enumerate(r.is_rc) if f] == [2, 5]


def test_extension_mode_on_ragged_reads():      # reference -m 2 on reads cut at random places (goldens from the reference CLI; the GPU twin is in test_gpu_device_general.py)
    fa = os.path.join(D, "out_ragged_ext_cons", "input.fa")
    assert _run_fx(fa, api.Params(aln_mode=2), True, False)[0] == _golden("out_ragged_ext_cons")
    assert _run_fx(fa, api.Params(aln_mode=2), False, True)[0] == _golden("out_ragged_ext_msa")


def test_quality_weights():                     # reference -Q, src/abpoa_align.c:462-467, abpoa_graph.c:486-499 / :634-667
    fq = os.path.join(D, "out_qv_cons", "input.fq")
    txt, _ = _run_fx(fq, api.Params(**AG), True, False, qv=True)
    assert txt == _golden("out_qv_cons")
    txt, _ = _run_fx(fq, api.Params(), True, True, qv=True)
    assert txt == _golden("out_qv_msa")


def test_pyabpoa_readme_example():              # reference python/README.md:28-33 (golden captured via pyabpoa)
    seqs = ["CCGAAGA", "CCGAACTCGA", "CCCGGAAGA", "CCGAAGA"]
    r = api.msa_batch([seqs], api.Params(), out_cons=True, out_msa=True, lib=H.cpu_shim_lib())[0]
    assert r.cons_seq == "CCGAAGA" and r.cons_cov == [4] * 7
    assert r.msa_seq == ["CC--GAA---GA", "CC--GAACTCGA", "CCCGGAA---GA", "CC--GAA---GA", "CC--GAA---GA"]


def test_ragged_batch_matches_single_runs():
    """Sets with different read counts in one lock-step batch == each set run alone."""
    sets = [synth.make_read_set(21, i, n, 120, 0.08) for i, n in enumerate((1, 2, 5, 3, 7))]
    p = api.Params(**AG)
    lib = H.cpu_shim_lib()
    together = api.msa_batch(sets, p, out_cons=True, out_msa=True, lib=lib, n_threads=3)
    for s, t in zip(sets, together):
        alone = api.msa_batch([s], p, out_cons=True, out_msa=True, lib=lib, n_threads=1)[0]
        assert t.cons_seq == alone.cons_seq and t.msa_seq == alone.msa_seq and t.cons_cov == alone.cons_cov
    assert together[0].cons_seq == sets[0][0]     # a single read is its own consensus


@pytest.mark.skipif(not H.have_ref(), reason="reference build not available")
@pytest.mark.parametrize("opts,pk,kw", [
    (["-O", "4,0", "-E", "2"], AG, {}),
    ([], {}, {}),
    (["-O", "0,0", "-E", "2"], dict(gap_open1=0, gap_open2=0), {}),
    (["-r", "2"], {}, dict(out_cons=True, out_msa=True)),
    (["-m", "1", "-r", "1"], dict(aln_mode=api.LOCAL), dict(out_cons=False, out_msa=True)),
    (["-m", "2"], dict(aln_mode=api.EXTEND), {}),
    (["-b", "-1"], dict(extra_b=-1), {}),
])
def test_against_reference_cli(tmp_path, opts, pk, kw):
    """Whole pipeline vs the compiled reference binary on seeded synthetic read-sets."""
    for seed, n, L, err in ((3, 12, 300, 0.1), (4, 20, 150, 0.2)):
        reads = synth.make_read_set(seed, 0, n, L, err)
        fa = str(tmp_path / f"s{seed}.fa")
        synth.write_fasta(fa, reads)
        exp = subprocess.run([os.path.join(H.REF_DIR, "abpoa_ref")] + opts + [fa], capture_output=True, text=True, check=True).stdout
        names = [f"r{i}" for i in range(n)]
        r = api.msa_batch([reads], api.Params(**pk), lib=H.cpu_shim_lib(), **kw)[0]
        assert api.format_output(r, names, kw.get("out_cons", True), kw.get("out_msa", False)) == exp
