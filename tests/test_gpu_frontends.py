"""GPU (-m gpu): the pyabpoa-compatible class and the command line on the real engine (device-resident driver where eligible)."""
import io
import os

import pytest

import helpers as H

pytestmark = pytest.mark.gpu
D = H.GOLDEN_DIR


def _golden(name):
    return open(os.path.join(D, name, "output.txt")).read()


@pytest.fixture(scope="module", autouse=True)
def engine():
    from abpoa_amd import ffi
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1
    ffi.check(lib.abpoa_hip_init(0))


def test_cli_list_mode_on_gpu(tmp_path):
    from abpoa_amd import cli
    seq = os.path.join(D, "data", "seq.fa"); s1k = os.path.join(D, "out_s1k_cons", "input.fa")
    lst = tmp_path / "list.txt"
    lst.write_text(seq + "\n" + s1k + "\n")
    buf = io.StringIO()
    assert cli.main(["-O", "4,0", "-E", "2", "-l", str(lst)], out=buf) == 0
    assert buf.getvalue() == _golden("out_seq_cons") + _golden("out_s1k_cons")
    buf = io.StringIO()
    assert cli.main(["-r", "2", os.path.join(D, "data", "test.fa")], out=buf) == 0       # MSA output: host driver
    assert buf.getvalue() == _golden("out_test_cons_msa")


def test_pyabpoa_class_on_gpu():
    from abpoa_amd import pyabpoa
    r = pyabpoa.msa_aligner().msa(["CCGAAGA", "CCGAACTCGA", "CCCGGAAGA", "CCGAAGA"], out_cons=True, out_msa=True)
    assert r.cons_seq == ["CCGAAGA"] and r.cons_cov == [[4] * 7]
    assert r.msa_seq == ["CC--GAA---GA", "CC--GAACTCGA", "CCCGGAA---GA", "CC--GAA---GA", "CC--GAA---GA"]
    r = pyabpoa.msa_aligner().msa(["CCGAAGA", "CCGAACTCGA", "CCCGGAAGA", "CCGAAGA"], out_cons=True, out_msa=False)   # device-resident driver
    assert r.cons_seq == ["CCGAAGA"] and r.cons_cov == [[4] * 7] and r.msa_seq == []


def test_pyabpoa_matches_the_reference_module_on_gpu():
    """The fixtures recorded from the reference's cythonized pyabpoa (oracle/make_pyabpoa_golden.py), through the engine; plus the CLI's -s / -Q."""
    import json
    from abpoa_amd import cli, pyabpoa
    for case in json.load(open(os.path.join(D, "pyabpoa", "cases.json")))["cases"]:
        r = pyabpoa.msa_aligner(**case["ctor"]).msa(case["seqs"], **case["msa"])
        got = dict(n_seq=r.n_seq, n_cons=r.n_cons, clu_n_seq=r.clu_n_seq, clu_read_ids=r.clu_read_ids, cons_len=r.cons_len, cons_seq=r.cons_seq,
                   cons_cov=r.cons_cov, msa_len=r.msa_len, msa_seq=r.msa_seq)
        assert got == case["expect"], case["name"]
    buf = io.StringIO()
    assert cli.main(["-s", "-r", "2", os.path.join(D, "out_rc_cons", "input.fa")], out=buf) == 0
    assert buf.getvalue() == _golden("out_rc_msa")
    buf = io.StringIO()
    assert cli.main(["-O", "4,0", "-E", "2", "-Q", os.path.join(D, "out_qv_cons", "input.fq")], out=buf) == 0
    assert buf.getvalue() == _golden("out_qv_cons")
