"""CPU: the two user-facing front ends over the read-set API -- the pyabpoa-compatible class (abpoa_amd/pyabpoa.py) and the
command line with the reference's option letters (abpoa_amd/cli.py) -- exercised through the oracle-backed host build
(tests/_build/libcpu_shim.so).  Expected values: the reference's printed outputs under tests/golden/out_* and the values
captured from the reference's own pyabpoa build (SURVEY.md 8c i, iii)."""
import io
import os

import pytest

import helpers as H
from abpoa_amd import cli, pyabpoa

D = H.GOLDEN_DIR


def _golden(name):
    return open(os.path.join(D, name, "output.txt")).read()


def test_pyabpoa_readme_example():              # reference python/README.md:28-33
    a = pyabpoa.msa_aligner(_lib=H.cpu_shim_lib())
    r = a.msa(["CCGAAGA", "CCGAACTCGA", "CCCGGAAGA", "CCGAAGA"], out_cons=True, out_msa=True)
    assert (r.n_seq, r.n_cons, r.clu_n_seq, r.cons_len) == (4, 1, [4], [7])
    assert r.cons_seq == ["CCGAAGA"] and r.cons_cov == [[4] * 7] and r.clu_read_ids == [[0, 1, 2, 3]]
    assert r.msa_len == 12 and r.msa_seq == ["CC--GAA---GA", "CC--GAACTCGA", "CCCGGAA---GA", "CC--GAA---GA", "CC--GAA---GA"]


def test_pyabpoa_seq_fa_affine():               # msa_aligner(aln_mode='g', gap_open2=0) on test_data/seq.fa
    from abpoa_amd import seqio
    _, seqs = seqio.read_fasta(os.path.join(D, "data", "seq.fa"))
    r = pyabpoa.msa_aligner(aln_mode='g', gap_open2=0, _lib=H.cpu_shim_lib()).msa(seqs, True, True)
    assert r.cons_seq == ["CGTCAATCTATCGAAGCATACGCGGCAGAGCCGAAGACCTCGGCAATCAC"] and r.cons_len == [50] and r.msa_len == 75
    assert r.cons_cov[0][:10] == [10, 10, 10, 10, 10, 10, 10, 9, 9, 10]
    r2 = pyabpoa.msa_aligner(aln_mode='g', gap_open2=0, _lib=H.cpu_shim_lib()).msa(seqs, True, False)
    assert r2.msa_seq == [] and r2.msa_len == 0 and r2.cons_seq == r.cons_seq


def _pyabpoa_cases():
    import json
    return json.load(open(os.path.join(D, "pyabpoa", "cases.json")))["cases"]


@pytest.mark.parametrize("case", _pyabpoa_cases(), ids=lambda c: c["name"])
def test_pyabpoa_matches_the_reference_module(case):
    """Fixtures recorded from the reference's own cythonized pyabpoa (oracle/make_pyabpoa_golden.py): every result attribute."""
    a = pyabpoa.msa_aligner(_lib=H.cpu_shim_lib(), **case["ctor"])
    r = a.msa(case["seqs"], **case["msa"])
    e = case["expect"]
    got = dict(n_seq=r.n_seq, n_cons=r.n_cons, clu_n_seq=r.clu_n_seq, clu_read_ids=r.clu_read_ids, cons_len=r.cons_len, cons_seq=r.cons_seq,
               cons_cov=r.cons_cov, msa_len=r.msa_len, msa_seq=r.msa_seq)
    assert got == e


def test_pyabpoa_rejects_what_the_engine_does_not_build():
    a = pyabpoa.msa_aligner(_lib=H.cpu_shim_lib())
    with pytest.raises(NotImplementedError):
        a.msa(["ACGT", "ACGA"], True, False, max_n_cons=2)
    with pytest.raises(Exception):
        pyabpoa.msa_aligner(aln_mode='x')


def test_pyabpoa_batch_equals_single_calls():
    from abpoa_amd import synth
    sets = [synth.make_read_set(5, i, 6, 90, 0.08) for i in range(4)]
    a = pyabpoa.msa_aligner(gap_open2=0, _lib=H.cpu_shim_lib())
    together = a.msa_batch(sets, out_cons=True, out_msa=True)
    for s, t in zip(sets, together):
        one = a.msa(s, True, True)
        assert (t.cons_seq, t.cons_cov, t.msa_seq) == (one.cons_seq, one.cons_cov, one.msa_seq)


def _cli(args):
    buf = io.StringIO()
    rc = cli.main(args, lib=H.cpu_shim_lib(), out=buf)
    assert rc == 0
    return buf.getvalue()


def test_cli_matches_reference_outputs():
    seq = os.path.join(D, "data", "seq.fa"); test = os.path.join(D, "data", "test.fa"); heter = os.path.join(D, "data", "heter.fa")
    assert _cli(["-O", "4,0", "-E", "2", seq]) == _golden("out_seq_cons")            # BASELINE.json config 1
    assert _cli(["-r", "1", test]) == _golden("out_test_msa")
    assert _cli(["-r", "2", test]) == _golden("out_test_cons_msa")
    assert _cli([heter]) == _golden("out_heter_cons")
    assert _cli(["-O", "4,0", "-E", "2", "-r", "5", seq]) == _golden("out_fq_seq")      # the consensus as FASTQ: a quality per base from its coverage (reference src/abpoa_output.c:270-276)
    assert _cli(["-r", "5", heter]) == _golden("out_fq_heter")
    assert _cli(["-m", "1", "-c", "-t", os.path.join(D, "data", "BLOSUM62.mtx"), "-r", "1", os.path.join(D, "aa_blosum_loc", "input.fa")]) == _golden("aa_blosum_loc")


def test_cli_list_mode_is_one_batch(tmp_path):
    seq = os.path.join(D, "data", "seq.fa"); s1k = os.path.join(D, "out_s1k_cons", "input.fa")
    lst = tmp_path / "list.txt"
    lst.write_text(seq + "\n" + s1k + "\n")
    assert _cli(["-O", "4,0", "-E", "2", "-l", str(lst)]) == _golden("out_seq_cons") + _golden("out_s1k_cons")


def test_cli_refuses_unsupported_output_modes():
    assert cli.main(["-r", "3", os.path.join(D, "data", "seq.fa")], lib=H.cpu_shim_lib(), out=io.StringIO()) == 2


def test_pyabpoa_set_seq_int_dict():
    """The binding's module-level helper (reference python/pyabpoa.pyx:69-86): codes of both cases, U as T, defaults for unknown letters / codes."""
    from abpoa_amd import pyabpoa as pa
    s2i, i2s = pa.set_seq_int_dict(5)
    assert [s2i[c] for c in "ACGTNacgtnUu"] == [0, 1, 2, 3, 4, 0, 1, 2, 3, 4, 3, 3] and s2i["X"] == 4 and i2s[3] == "T" and i2s[9] == "-"
    s2i, i2s = pa.set_seq_int_dict(27)
    assert "".join(i2s[i] for i in range(27)) == "ACGTNBDEFHIJKLMOPQRSUVWXYZ*" and s2i["w"] == s2i["W"] == 22 and s2i["?"] == 26 and i2s[30] == "-"
    import pytest
    with pytest.raises(Exception):
        pa.set_seq_int_dict(4)
    assert bool(pa.msa_aligner(_lib=object()))
