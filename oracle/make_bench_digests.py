#!/usr/bin/env python3
"""TEST INFRASTRUCTURE -- per-set reference outputs for the workloads bench.py measures.

Runs the compiled, unmodified reference (oracle/_ref/abpoa_ref, built by oracle/Makefile from /root/reference with
gcc -O3 -mavx2 -fno-strict-aliasing) over the very read-sets bench.py generates (abpoa_amd.synth, seed 1, set index =
rank * sets_per_gpu + i) and stores the sha256 of each set's output text:

    tests/golden/bench_digests/<workload>.json = {"workload", "seed", "options", "n_sets", "sha256": [per set ...]}

bench.py compares EVERY set of a run against these lists (parity_sets_checked), the -m gpu tests compare full-size sets.
The hashed text is exactly what `abpoa_ref <options> set.fa` prints (consensus FASTA, or the MSA for cfg5).
Only runs where /root/reference was available to build oracle/_ref; the lists it writes are data.

usage: python oracle/make_bench_digests.py [workload ...]     (default: all)
"""
import hashlib
import json
import os
import subprocess
import sys
import tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abpoa_amd import synth  # noqa: E402
from abpoa_amd.workloads import WORKLOADS, DIGEST_SETS, ref_options  # noqa: E402

REF_BIN = os.path.join(ROOT, "oracle", "_ref", "abpoa_ref")
OUT_DIR = os.path.join(ROOT, "tests", "golden", "bench_digests")


def one(wl, idx, tmp):
    w = WORKLOADS[wl]
    reads = synth.make_read_set(1, idx, **synth.CONFIGS[w["cfg"]])
    fn = os.path.join(tmp, f"{wl}_{idx}.fa")
    synth.write_fasta(fn, reads)
    env = dict(os.environ, GLIBC_TUNABLES="glibc.malloc.hugetlb=1")
    txt = subprocess.run([REF_BIN] + ref_options(wl) + [fn], capture_output=True, check=True, env=env).stdout
    os.unlink(fn)
    return hashlib.sha256(txt).hexdigest()


def main():
    names = sys.argv[1:] or sorted(DIGEST_SETS)
    os.makedirs(OUT_DIR, exist_ok=True)
    for wl in names:
        n = DIGEST_SETS[wl]
        have = []      # (a list that is being extended keeps the digests it has: set i's reads depend on (seed, i) only)
        fn_old = os.path.join(OUT_DIR, wl + ".json")
        if os.path.exists(fn_old):
            old = json.load(open(fn_old))
            if old.get("seed") == 1 and old.get("options") == ref_options(wl, portable=True):
                have = old["sha256"][:n]
        with tempfile.TemporaryDirectory(prefix="abpoa_dig_") as tmp, ThreadPoolExecutor(max_workers=2 if wl == "cfg3l" else 7) as ex:      # (the reference holds rows x (qlen + 1) x planes: ~25 GB per process on 50 x 20 kb reads)
            shas = have + list(ex.map(lambda i: one(wl, i, tmp), range(len(have), n)))
        rec = {"workload": wl, "seed": 1, "options": ref_options(wl, portable=True), "n_sets": n,
               "generator": "oracle/make_bench_digests.py (abPOA v1.4.1, oracle/_ref/abpoa_ref)", "sha256": shas}
        with open(os.path.join(OUT_DIR, wl + ".json"), "w") as f:
            json.dump(rec, f, indent=0)
        print(wl, n, "sets", hashlib.sha256("".join(shas).encode()).hexdigest()[:16], flush=True)


if __name__ == "__main__":
    main()
