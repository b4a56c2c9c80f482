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


@pytest.mark.skipif(not H.have_ref(), reason="reference build not available")
def test_seeded_sweep_of_the_oracle_backed_host_layer_against_the_reference_cli(tmp_path):
    """What the GPU fuzz sweeps compare the device with -- the CPU build of the host layer with the oracle as its aligner -- against the compiled reference
    binary on the same kind of option / shape mixes (tools/fuzz_device_vs_oracle.py: nucleotides or amino acids, linear / affine / convex gaps, global /
    local / extension mode, band on / off / narrow, z-drop, -s), output text byte for byte (`-r 2`: MSA rows and consensus).  An input on which the
    reference itself fails (its backtrack after a z-drop break) is skipped and counted."""
    import numpy as np
    from abpoa_amd import workloads
    ref_bin = os.path.join(H.REF_DIR, "abpoa_ref")
    comp = str.maketrans("ACGTN", "TGCAN")
    n_ok = n_ref_failed = 0
    for it in range(160):
        rng = np.random.default_rng(9100 + it)
        aa = rng.random() < 0.2
        n = int(rng.integers(2, 26)); ln = int(rng.integers(40, 700)); err = float(rng.uniform(0.01, 0.15))
        reads = list(synth.make_read_set(9100 + it, 0, n, ln, alphabet=synth.AA, rates=(err, err / 3, err / 3))) if aa else list(synth.make_read_set(9100 + it, 0, n, ln, err))
        if rng.random() < 0.4:      # ragged ends
            reads = [reads[0]] + [r[int(rng.integers(0, len(r) // 8 + 1)):len(r) - int(rng.integers(0, len(r) // 8 + 1))] for r in reads[1:]]
        gap = [dict(gap_open1=0, gap_open2=0, gap_ext1=int(rng.integers(1, 5))), dict(gap_open1=int(rng.integers(2, 12)), gap_open2=0, gap_ext1=int(rng.integers(1, 4))), dict(),
               dict(gap_open1=int(rng.integers(3, 8)), gap_open2=int(rng.integers(12, 40)), gap_ext1=int(rng.integers(2, 4)), gap_ext2=1)][int(rng.integers(0, 4))]
        mode = int(rng.integers(0, 3))
        kw = dict(gap, aln_mode=mode)
        if mode != 1:
            r_ = rng.random()
            if r_ < 0.2:
                kw["extra_b"] = -1
            elif r_ < 0.5:
                kw.update(extra_b=int(rng.integers(2, 60)), extra_f=float(rng.choice([0.0, 0.01, 0.03])))
        if mode == 2 and rng.random() < 0.4:
            kw["zdrop"] = int(rng.integers(10, 100))
        amb = (not aa) and rng.random() < 0.3
        if amb:
            reads = [(r[::-1].translate(comp) if (j and rng.random() < 0.3) else r) for j, r in enumerate(reads)]
        p = api.Params(is_aa=True, score_matrix=workloads.BLOSUM62, **kw) if aa else api.Params(**kw)
        opts = ["-m", str(mode), "-O", f"{p.gap_open1},{p.gap_open2}", "-E", f"{p.gap_ext1},{p.gap_ext2}", "-b", str(kw.get("extra_b", 10)), "-f", str(kw.get("extra_f", 0.01)), "-r", "2"]
        if "zdrop" in kw:
            opts += ["-z", str(kw["zdrop"])]
        if aa:
            opts += ["-c", "-t", workloads.BLOSUM62]
        if amb:
            opts += ["-s"]
        fa = str(tmp_path / f"f{it}.fa")
        synth.write_fasta(fa, reads)
        ref = subprocess.run([ref_bin] + opts + [fa], capture_output=True, text=True, timeout=300)
        r = api.msa_batch([reads], p, out_cons=True, out_msa=True, lib=H.cpu_shim_lib(), n_threads=2, amb_strand=amb)[0]
        if ref.returncode != 0:
            n_ref_failed += 1
            assert r.status != 0, (it, opts, "the reference fails on this input, the host layer does not")
            continue
        assert r.status == 0, (it, opts, r.status)
        assert api.format_output(r, [f"r{i}" for i in range(len(reads))], True, True) == ref.stdout, (it, opts, [len(x) for x in reads])
        n_ok += 1
    print(f"sweep: {n_ok} outputs identical to abpoa_ref, {n_ref_failed} inputs on which the reference itself fails")
    assert n_ok >= 140 and n_ref_failed <= 20, (n_ok, n_ref_failed)
