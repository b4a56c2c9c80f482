#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- regenerates tests/golden/ from the compiled reference.

Runs oracle/_ref/ref_dump (the unmodified abPOA v1.4.1, built by oracle/Makefile with
gcc -O3 -mavx2 -fno-strict-aliasing) over the reference's own test data and over seeded synthetic
read-sets, and stores per-alignment containers (inputs + the reference's bands, planes or per-row
plane checksums, best score, cigar) as tests/golden/<case>/aln_XXX.abpg.gz plus the reference's final
consensus / MSA text.  Only runs where /root/reference exists; the fixtures it writes are data.

usage: python oracle/make_golden.py
"""
import gzip
import glob
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402
from abpoa_amd import synth  # noqa: E402

REF = H.REFERENCE_TREE
AG = ["-O", "4,0", "-E", "2"]
LG = ["-O", "0,0", "-E", "2"]
CG = []
MODES = {"gb": [], "gu": ["-b", "-1"], "loc": ["-m", "1"], "ext": ["-m", "2"], "extz": ["-m", "2", "-z", "5"]}


def cases(tmp):
    s1k = os.path.join(tmp, "s1k.fa")
    synth.write_fasta(s1k, synth.make_read_set(1, 0, 12, 1000, 0.05))
    s10k = os.path.join(tmp, "s10k.fa")
    synth.write_fasta(s10k, synth.make_read_set(2, 0, 9, 10000, 0.15))
    aa = os.path.join(tmp, "aa.fa")
    synth.write_fasta(aa, synth.make_read_set(5, 0, 8, 500, alphabet=synth.AA, rates=(0.05, 0.03, 0.03)))
    seq, het, tst = (os.path.join(REF, "test_data", f) for f in ("seq.fa", "heter.fa", "test.fa"))
    out = []
    for gname, g in (("ag", AG), ("cg", CG), ("lg", LG)):
        for mname, mo in MODES.items():
            out.append((f"seq_{gname}_{mname}", seq, g + mo, "1,5,9", 1, None))
        out.append((f"test_{gname}_gb", tst, g, "all", 1, None))
        out.append((f"heter_{gname}_gb", het, g, "7", 0, None))
    out.append(("heter_cg_extz", het, CG + MODES["extz"], "5,12", 0, None))
    out.append(("seq_ag_sub", seq, AG, "3,8", 1, (10, 40)))
    out.append(("heter_cg_sub", het, CG, "9", 0, (100, 500)))
    out.append(("s1k_ag_gb", s1k, AG, "11", 0, None))
    out.append(("s1k_cg_gb", s1k, CG, "5", 0, None))
    out.append(("s10k_cg_i32", s10k, CG, "8", 0, None))
    out.append(("s10k_ag_i32", s10k, AG, "8", 0, None))
    s20k = os.path.join(tmp, "s20k.fa")                 # reads above 16 K bases: query beyond the old LDS limit of the fast row loops, band of ~440 columns
    synth.write_fasta(s20k, synth.make_read_set(3, 0, 4, 20000, 0.05))
    out.append(("s20k_ag_i32", s20k, AG, "3", 0, None))
    out.append(("aa_blosum_loc", aa, ["-m", "1", "-c", "-t", os.path.join(REF, "BLOSUM62.mtx"), "-r", "1"], "3,7", 0, None))
    out.append(("aa_blosum_gb", aa, ["-c", "-t", os.path.join(REF, "BLOSUM62.mtx")], "7", 0, None))
    # local alignment of reads with ragged ends (every read but the first a random substring of its noisy full-length version): read 9's best path enters the
    # graph through a node 24 rows above its successor -- the backtrack leaves its staged window for a slow step there (the case that showed a stale
    # window record in the tail kernel's local walk: tests/test_gpu_device_general.py::test_reads_with_ragged_ends...)
    import numpy as np
    rg = os.path.join(tmp, "ragged.fa")
    rng = np.random.default_rng(109)
    full = synth.make_read_set(109, 0, 45, 260, 0.04)
    cut = [full[0]]
    for r in full[1:]:
        a = int(rng.integers(0, int(0.15 * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(0.15 * len(r)) + 1))
        cut.append(r[a:b])
    synth.write_fasta(rg, cut[:10])
    out.append(("ragged_cg_loc", rg, ["-m", "1"], "5,9", 0, None))
    out.append(("ragged_ag_loc", rg, AG + ["-m", "1"], "9", 0, None))
    # whole-pipeline text goldens (consensus / MSA) used by the host-layer tests
    out.append(("out_seq_cons", seq, AG, "none", 0, None))
    out.append(("out_test_msa", tst, ["-r", "1"], "none", 0, None))
    out.append(("out_test_cons_msa", tst, ["-r", "2"], "none", 0, None))
    out.append(("out_heter_cons", het, CG, "none", 0, None))
    out.append(("out_s1k_cons", s1k, AG, "none", 0, None))
    # -s (ambiguous strand): reads 2, 5, 8 of a seeded set handed over as their reverse complement; -Q: FASTQ with seeded qualities
    rcfa = os.path.join(tmp, "rc.fa")
    rs = synth.make_read_set(7, 0, 10, 400, 0.08)
    comp = str.maketrans("ACGT", "TGCA")
    synth.write_fasta(rcfa, [r[::-1].translate(comp) if i in (2, 5, 8) else r for i, r in enumerate(rs)])
    out.append(("out_rc_cons", rcfa, AG + ["-s"], "none", 0, None))
    out.append(("out_rc_msa", rcfa, ["-s", "-r", "2"], "none", 0, None))
    # ... and on long noisy reads under the default adaptive band (-b 10 -f 0.01): the retry starts from the band bounds the forward DP left behind
    rcl = os.path.join(tmp, "rc_long.fa")
    rl = synth.make_read_set(21, 0, 8, 2500, 0.15)
    synth.write_fasta(rcl, [r[::-1].translate(comp) if i in (2, 5) else r for i, r in enumerate(rl)])
    out.append(("out_rc_long_msa", rcl, ["-s", "-r", "2"], "none", 0, None))
    # global alignment of a short read against a longer graph (set 98 of bench.py's ragged entry): in the wide row loop with a 12 KB backtrack window the
    # int16-affine records' slices (two columns wider than planned: even-column rounding) overran the window -- EBACKTRACK, hidden by the retry passes
    rng98 = np.random.default_rng(20240)
    for i98 in range(99):
        full98 = synth.make_read_set(1, i98, **synth.CONFIGS[2])
        cut98 = [full98[0]]
        for r in full98[1:]:
            a = int(rng98.integers(0, len(r) // 10 + 1)); b = len(r) - int(rng98.integers(0, len(r) // 10 + 1))
            cut98.append(r[a:b])
    r98 = os.path.join(tmp, "ragged98.fa")
    synth.write_fasta(r98, cut98[:6])
    out.append(("ragged_ag_gb", r98, AG, "1,4", 0, None))
    # extension mode on ragged reads: read 6 of this set finds no alignment to speak of (best score 2: one base on a successor of the source 800 rows down the
    # order) -- the general kernel's band state of far successors of the source (tools/fuzz_device_vs_oracle.py, seed 100049, set 3)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import numpy as np
    import fuzz_device_vs_oracle as fz
    frng = np.random.default_rng(100049)
    faa = frng.random() < 0.2
    rext = os.path.join(tmp, "ragged_ext.fa")
    synth.write_fasta(rext, fz.make_sets(frng, 100049, faa)[3])
    out.append(("out_ragged_ext_cons", rext, ["-m", "2"], "none", 0, None))
    out.append(("out_ragged_ext_msa", rext, ["-m", "2", "-r", "1"], "none", 0, None))
    qfq = os.path.join(tmp, "qv.fq")
    rng = synth.SplitMix64(99)
    with open(qfq, "w") as f:
        for i, r in enumerate(synth.make_read_set(8, 0, 9, 300, 0.1)):
            q = "".join(chr(33 + int(x % 41)) for x in rng.block(len(r)))
            f.write(f"@q{i}\n{r}\n+\n{q}\n")
    out.append(("out_qv_cons", qfq, AG + ["-Q"], "none", 0, None))
    out.append(("out_qv_msa", qfq, ["-Q", "-r", "2"], "none", 0, None))
    # -r 5: the consensus as FASTQ (a quality per base from its coverage, reference src/abpoa_output.c:270-276)
    out.append(("out_fq_seq", seq, AG + ["-r", "5"], "none", 0, None))
    out.append(("out_fq_heter", het, ["-r", "5"], "none", 0, None))
    return out


ONLY = set(sys.argv[1:])      # (names given on the command line: only those containers are rewritten)


def main():
    if not H.have_ref():
        sys.exit("oracle/_ref/ref_dump missing: run `make -C oracle` where /root/reference exists")
    tmp = tempfile.mkdtemp()
    for name, fa, opts, reads, planes, sub in cases(tmp):
        d = os.path.join(tmp, name)
        if ONLY and name not in ONLY:
            continue
        if name.startswith(("out_rc_", "out_qv_", "out_fq_")):      # options the dump harness does not parse: the reference's own command line prints the text
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "output.txt"), "w") as fo:
                subprocess.run([os.path.join(H.REF_DIR, "abpoa_ref")] + opts + [fa], stdout=fo, check=True)
        else:
            H.run_ref_dump(fa, d, opts, reads, planes=planes, sub=sub)
        dst = os.path.join(H.GOLDEN_DIR, name)
        shutil.rmtree(dst, ignore_errors=True)
        os.makedirs(dst)
        for f in sorted(glob.glob(os.path.join(d, "*.abpg"))):
            with open(f, "rb") as fi, gzip.GzipFile(os.path.join(dst, os.path.basename(f) + ".gz"), "wb", mtime=0) as fo:
                fo.write(fi.read())
        shutil.copy(os.path.join(d, "output.txt"), os.path.join(dst, "output.txt"))
        with open(os.path.join(dst, "cmd.txt"), "w") as f:
            f.write(" ".join(o.replace(REF, "$REF").replace(tmp, "$TMP") for o in opts) + f" | reads={reads} sub={sub} input={os.path.basename(fa)}\n")
        if fa.startswith(tmp):
            shutil.copy(fa, os.path.join(dst, "input.fq" if fa.endswith(".fq") else "input.fa"))
    # the reference's own small inputs and matrices are data fixtures too
    fx = os.path.join(H.GOLDEN_DIR, "data")
    os.makedirs(fx, exist_ok=True)
    for f in ("test_data/seq.fa", "test_data/heter.fa", "test_data/test.fa", "BLOSUM62.mtx", "HOXD70.mtx", "PAM250.mtx"):
        shutil.copy(os.path.join(REF, f), os.path.join(fx, os.path.basename(f)))
    shutil.rmtree(tmp)
    print("golden fixtures written to", H.GOLDEN_DIR)


if __name__ == "__main__":
    main()
