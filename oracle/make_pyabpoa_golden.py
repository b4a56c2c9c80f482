#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- goldens for the pyabpoa-compatible front end (abpoa_amd/pyabpoa.py), captured from the reference's OWN Python
module.

The reference's python/pyabpoa.pyx is cythonized and built against the reference's C sources where they lie under /root/reference (same
flags as oracle/Makefile: gcc -O3 -mavx2 -fno-strict-aliasing, no -DUSE_SIMDE) in a temporary directory OUTSIDE the repository, which is
removed as soon as the fixtures are written: neither the generated C nor the compiled module ever sits in the tree (so it cannot travel to
a GPU box with a snapshot of it).  The module is imported here only to record inputs -> outputs as JSON fixtures under
tests/golden/pyabpoa/, which the CPU and GPU front-end tests read.  Only runs where /root/reference exists.

usage: python oracle/make_pyabpoa_golden.py
"""
import glob
import json
import os
import shutil
import subprocess
import sys
import sysconfig
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
sys.path.insert(0, ROOT)


def build(OUT):
    c_file = os.path.join(OUT, "pyabpoa.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-3", "-I", os.path.join(REF, "python"), os.path.join(REF, "python", "pyabpoa.pyx"), "-o", c_file])
    srcs = [os.path.join(REF, "src", f + ".c") for f in ("abpoa_align", "abpoa_graph", "abpoa_output", "abpoa_plot", "abpoa_seed", "abpoa_seq", "kalloc",
                                                         "kstring", "simd_abpoa_align", "simd_check", "utils")]
    so = os.path.join(OUT, "pyabpoa" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O3", "-w", "-fPIC", "-shared", "-mavx2", "-fno-strict-aliasing", "-I" + os.path.join(REF, "include"), "-I" + os.path.join(REF, "src"),
                           "-I" + sysconfig.get_paths()["include"], c_file] + srcs + ["-o", so, "-lm", "-lz", "-lpthread"])
    return so


def record(res):
    return {"n_seq": res.n_seq, "n_cons": res.n_cons, "clu_n_seq": list(res.clu_n_seq), "clu_read_ids": [list(x) for x in res.clu_read_ids],
            "cons_len": list(res.cons_len), "cons_seq": list(res.cons_seq), "cons_cov": [list(x) for x in res.cons_cov],
            "msa_len": res.msa_len, "msa_seq": list(res.msa_seq)}


def main():
    OUT = tempfile.mkdtemp(prefix="pyabpoa_ref_")      # outside the repository
    try:
        build(OUT)
        sys.path.insert(0, OUT)
        capture()
    finally:
        shutil.rmtree(OUT, ignore_errors=True)


def capture():
    import pyabpoa as pa          # the REFERENCE's module
    from abpoa_amd import seqio, synth
    data = os.path.join(ROOT, "tests", "golden", "data")
    cases = []

    def add(name, seqs, ctor, msa_kw):
        a = pa.msa_aligner(**ctor)
        res = a.msa(seqs, **msa_kw)
        cases.append({"name": name, "seqs": seqs, "ctor": ctor, "msa": msa_kw, "expect": record(res)})

    add("readme", ["CCGAAGA", "CCGAACTCGA", "CCCGGAAGA", "CCGAAGA"], {}, dict(out_cons=True, out_msa=True))               # python/README.md:28-33
    _, seq_fa = seqio.read_fasta(os.path.join(data, "seq.fa"))
    add("seq_fa_affine", seq_fa, dict(aln_mode="g", gap_open2=0), dict(out_cons=True, out_msa=True))                           # BASELINE.json configs[0]
    add("seq_fa_affine_cons_only", seq_fa, dict(aln_mode="g", gap_open2=0), dict(out_cons=True, out_msa=False))
    _, het = seqio.read_fasta(os.path.join(data, "heter.fa"))
    add("heter_default", het, {}, dict(out_cons=True, out_msa=False))
    _, tst = seqio.read_fasta(os.path.join(data, "test.fa"))
    add("test_fa_msa", tst, {}, dict(out_cons=False, out_msa=True))
    add("synthetic_local", synth.make_read_set(31, 0, 8, 150, 0.1), dict(aln_mode="l"), dict(out_cons=True, out_msa=True))
    add("synthetic_linear", synth.make_read_set(32, 0, 8, 150, 0.1), dict(gap_open1=0, gap_open2=0), dict(out_cons=True, out_msa=True))
    add("synthetic_unbanded", synth.make_read_set(33, 0, 8, 150, 0.1), dict(extra_b=-1), dict(out_cons=True, out_msa=True))
    dst = os.path.join(ROOT, "tests", "golden", "pyabpoa")
    os.makedirs(dst, exist_ok=True)
    with open(os.path.join(dst, "cases.json"), "w") as f:
        json.dump({"generator": "oracle/make_pyabpoa_golden.py (reference pyabpoa 1.4.1, cythonized here)", "cases": cases}, f, indent=1)
    print(len(cases), "cases written to", dst)


if __name__ == "__main__":
    main()
