#!/usr/bin/env python3
"""Reads of 20 kb (band half-width w = 10 + 0.01 L = 210: rows of 7 - 9 chunks, wider than the 448-column ring): the long-read form of the wide row loop
(dp_xl_rows.hip) against the chunk-by-chunk bodies such rows took before (ABPOA_HIP_NOXL=1), same consensus required.  usage: python tools/long_reads_bench.py
[--sets 64] [--reads 20] [--len 20000] [--err 0.1]"""
import argparse
import json
import os
import subprocess
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(a):
    from abpoa_amd import api, ffi, synth
    lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
    sets = [synth.make_read_set(7, i, a.reads, a.len, a.err) for i in range(a.sets)]
    p = api.Params()          # convex defaults, -b 10 -f 0.01
    enc = api.EncodedSets(sets, p.m)
    api.msa_batch(None, p, encoded=enc, n_threads=8)
    t0 = time.perf_counter(); res = api.msa_batch(None, p, encoded=enc, n_threads=8); dt = time.perf_counter() - t0
    assert all(r.status == 0 for r in res)
    import hashlib
    h = hashlib.sha256("\n".join(r.cons_seq for r in res).encode()).hexdigest()
    print(json.dumps({"xl": not os.environ.get("ABPOA_HIP_NOXL"), "sets": a.sets, "reads": a.reads, "len": a.len, "sets_per_s": round(a.sets / dt, 3), "s": round(dt, 3),
                      "n_host_sets": api.msa_timing()["n_host_sets"], "consensus_sha256": h}), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sets", type=int, default=64); ap.add_argument("--reads", type=int, default=20); ap.add_argument("--len", type=int, default=20000)
    ap.add_argument("--err", type=float, default=0.1); ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.child:
        return run(a)
    outs = []
    for env in ({}, {"ABPOA_HIP_NOXL": "1"}):
        p = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", "--sets", str(a.sets), "--reads", str(a.reads), "--len", str(a.len), "--err", str(a.err)],
                           capture_output=True, text=True, env=dict(os.environ, **env))
        if p.returncode != 0:
            raise SystemExit(p.stderr[-3000:])
        outs.append(json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]))
        print(outs[-1], flush=True)
    assert outs[0]["consensus_sha256"] == outs[1]["consensus_sha256"], "the two forms disagree"
    print(json.dumps({"long_reads": outs, "speedup": round(outs[0]["sets_per_s"] / outs[1]["sets_per_s"], 2)}))


if __name__ == "__main__":
    main()
