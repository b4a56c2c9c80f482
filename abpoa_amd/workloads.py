"""The BASELINE.json workloads in one place: generator arguments (abpoa_amd.synth.CONFIGS), engine parameters, the
reference command-line options of the same run and how many sets of each carry a committed
reference digest (tests/golden/bench_digests/, written by oracle/make_bench_digests.py).

bench.py, the digest generator and the GPU tests all read this table, so "the same workload" is one definition."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BLOSUM62 = os.path.join(ROOT, "tests", "golden", "data", "BLOSUM62.mtx")      # the reference's own matrix file (data)

WORKLOADS = {
    "cfg2": dict(cfg=2, params=dict(gap_open1=4, gap_open2=0, gap_ext1=2), ref_opts=["-O", "4,0", "-E", "2"], out_msa=False,
                 desc="50 reads x 1 kb, 5% err, global affine (-O 4,0 -E 2)"),
    "cfg3": dict(cfg=3, params=dict(), ref_opts=[], out_msa=False,
                 desc="50 reads x 10 kb, 15% err, global convex defaults (-b 10 -f 0.01)"),
    "cfg4": dict(cfg=4, params=dict(gap_open1=4, gap_open2=0, gap_ext1=2), ref_opts=["-O", "4,0", "-E", "2"], out_msa=False,
                 desc="50 reads x 10 kb, 5% err, global affine"),
    "cfg5": dict(cfg=5, params=dict(aln_mode=1, is_aa=True, score_matrix=BLOSUM62), ref_opts=["-m", "1", "-c", "-t", BLOSUM62, "-r", "1"],
                 out_msa=True, desc="30 seqs x 500 aa, local convex BLOSUM62 (-m 1 -c -t BLOSUM62.mtx -r 1), MSA output"),
    # not a BASELINE.json configuration: ONT reads of 20 kb (missing 4 -- the wide row loop ended at 448 columns)
    "cfg3l": dict(cfg=6, params=dict(), ref_opts=[], out_msa=False, desc="50 reads x 20 kb, 10% err, global convex defaults (-b 10 -f 0.01): long reads"),
}
# sets (index 0 .. n-1, seed 1) with a committed reference digest
DIGEST_SETS = {"cfg2": 8000, "cfg3": 2048, "cfg4": 2048, "cfg5": 1000, "cfg3l": 64}


def ref_options(wl, portable=False):
    o = list(WORKLOADS[wl]["ref_opts"])
    return [("BLOSUM62.mtx" if x == BLOSUM62 else x) for x in o] if portable else o


def load_digests(wl):
    fn = os.path.join(ROOT, "tests", "golden", "bench_digests", wl + ".json")
    if not os.path.exists(fn):
        return None
    return json.load(open(fn))["sha256"]


def output_sha(text):
    return hashlib.sha256(text.encode()).hexdigest()
