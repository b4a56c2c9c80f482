#!/usr/bin/env python3
"""BASELINE.json configs[3] from C, as a stream: `abpoa_amd/abpoa_batch -l` over N FASTA files of 50 x 10 kb reads -- the list is
read --piece files at a time by background threads while the GPU works on the piece before (abpoa_amd/host/abpoa_batch.c), so the job never exists as a whole
in memory.  Measures the end-to-end rate of the C front end (process start to last byte of output: file reading, parsing and encoding included), checks the
consensus of every file against the committed reference digests (tests/golden/bench_digests/cfg4.json, the first 2048 sets; every later file repeats one of
them) and, where oracle/_ref/abpoa_ref travels, byte-compares a 64-file sample with `abpoa_ref -l`.  Prints one JSON line.
usage: python tools/stream_cli_bench.py [--files 8192] [--piece 2048] [--readers 8] [--dir /tmp/stream_cli]"""
import argparse
import hashlib
import json
import multiprocessing as mp
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from abpoa_amd import synth, workloads      # noqa: E402


def _write(args):
    d, i, distinct = args
    reads = synth.make_read_set(1, i % distinct, **synth.CONFIGS[4])
    fn = os.path.join(d, f"s{i}.fa")
    with open(fn, "w") as f:
        f.write("".join(f">r{j}\n{r}\n" for j, r in enumerate(reads)))
    return fn


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--files", type=int, default=8192); ap.add_argument("--piece", type=int, default=2048); ap.add_argument("--readers", type=int, default=8)
    ap.add_argument("--dir", default="/tmp/stream_cli"); ap.add_argument("--distinct", type=int, default=2048, help="distinct read-sets (the files cycle through them)")
    ap.add_argument("--threads", type=int, default=0)
    a = ap.parse_args()
    os.makedirs(a.dir, exist_ok=True)
    t0 = time.time()
    with mp.get_context("fork").Pool(min(16, os.cpu_count() or 8)) as pool:
        files = pool.map(_write, [(a.dir, i, a.distinct) for i in range(a.files)], chunksize=16)
    lst = os.path.join(a.dir, "list.txt")
    open(lst, "w").write("\n".join(files) + "\n")
    gen_s = time.time() - t0
    total_bytes = sum(os.path.getsize(f) for f in files[:a.distinct]) * (a.files / min(a.files, a.distinct))
    exe = os.path.join(ROOT, "abpoa_amd", "abpoa_batch")
    opts = workloads.ref_options("cfg4")
    cmd = [exe] + opts + ["-l", lst, "--piece", str(a.piece), "--readers", str(a.readers)] + (["-T", str(a.threads)] if a.threads else [])
    t0 = time.time()
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(os.environ, ABPOA_BATCH_TIMING="1"))
    wall = time.time() - t0
    sys.stderr.write(p.stderr[-4000:])
    if p.returncode != 0:
        raise SystemExit(f"abpoa_batch failed ({p.returncode}): {p.stderr[-2000:]}")
    rss_gb = resource.getrusage(resource.RUSAGE_CHILDREN).ru_maxrss / 1e6
    recs = p.stdout.split(">Consensus_sequence\n")[1:]
    assert len(recs) == a.files, (len(recs), a.files)
    dig = workloads.load_digests("cfg4")
    bad = 0
    for i, r in enumerate(recs):
        h = hashlib.sha256((">Consensus_sequence\n" + r).encode()).hexdigest()
        if dig is not None and (i % a.distinct) < len(dig) and h != dig[i % a.distinct]:
            bad += 1
    out = {"what": "abpoa_batch -l (C front end, streamed list) on BASELINE.json configs[3] files: 50 reads x 10 kb, global affine", "files": a.files, "piece": a.piece, "readers": a.readers,
           "wall_s": round(wall, 2), "read_sets_per_s": round(a.files / wall, 1), "input_GB": round(total_bytes / 1e9, 2), "input_MB_per_s": round(total_bytes / wall / 1e6, 1),
           "peak_rss_GB_of_the_process": round(rss_gb, 2), "digest_mismatches": bad, "digests_checked": a.files if dig is not None else 0, "generate_files_s": round(gen_s, 1)}
    ref = os.path.join(ROOT, "oracle", "_ref", "abpoa_ref")
    if os.path.exists(ref):
        sample = os.path.join(a.dir, "sample.txt"); open(sample, "w").write("\n".join(files[:64]) + "\n")
        t0 = time.time(); r = subprocess.run([ref] + opts + ["-l", sample], capture_output=True, text=True); ref_s = time.time() - t0
        g = subprocess.run([exe] + opts + ["-l", sample, "--piece", "16"], capture_output=True, text=True)
        out["sample_64_files_byte_identical_to_abpoa_ref"] = (r.returncode == 0 and g.returncode == 0 and r.stdout == g.stdout)
        out["abpoa_ref_one_core_sets_per_s"] = round(64 / ref_s, 2)
    print(json.dumps(out), flush=True)
    if bad or out.get("sample_64_files_byte_identical_to_abpoa_ref") is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
