"""Seeded fuzzing of the device-resident driver against the CPU build of the host layer whose aligner is the plain-C oracle (tests/cpu_shim.cpp): random read-set
shapes (uncut / ragged ends / degenerate), nucleotide or protein, every gap model, alignment mode, band on / off, -s, per-base weights, consensus and MSA.
Prints the first mismatch with everything needed to replay it and exits 1.  usage: python tools/fuzz_device_vs_oracle.py [--iters N] [--seed S]"""
import argparse
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H                                   # noqa: E402
from abpoa_amd import api, ffi, synth, workloads     # noqa: E402

COMP = str.maketrans("ACGTN", "TGCAN")


def make_sets(rng, seed, aa, small=False):
    sets = []
    for i in range(int(rng.integers(2, 7))):
        kind = rng.random()
        u = rng.random()
        n = int(rng.integers(2, 60)); ln = int(rng.integers(30, 1400)) if u < 0.8 else (int(rng.integers(1400, 4000)) if u < 0.94 else int(rng.integers(5000, 12000)))
        if small:      # (the sweep inside the GPU test suite: the oracle-backed leg must stay at a fraction of a second per iteration)
            n = min(n, 20); ln = min(ln, 600)
        if ln > 1400:
            n = min(n, 14 if ln < 5000 else 6)
        err = float(rng.uniform(0.01, 0.15))
        if aa:
            reads = list(synth.make_read_set(seed, i, n, min(ln, 560), alphabet=synth.AA, rates=(err, err / 3, err / 3)))
        else:
            reads = list(synth.make_read_set(seed, i, n, ln, err))
        if kind < 0.45:                                # ragged ends
            frac = float(rng.uniform(0.03, 0.3)); out = [reads[0]]
            for r in reads[1:]:
                a = int(rng.integers(0, int(frac * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(frac * len(r)) + 1))
                out.append(r[a:max(a + 1, b)])
            reads = out
        elif kind < 0.55:                              # one read of a very different length
            j = int(rng.integers(0, len(reads))); reads[j] = reads[j][:max(1, len(reads[j]) // int(rng.integers(3, 12)))]
        elif kind < 0.6:                               # duplicates
            reads = reads[:2] * (len(reads) // 2 + 1)
        sets.append(reads)
    return sets


def iteration(seed, shim, small=False):
    """One seeded option / shape mix on the device-resident driver and on the oracle-backed host run.  Returns (sets on the device, sets through the host
    driver, sets with the same error status on both sides, {reason: sets} of the host-driver sets); raises AssertionError with a replay line on a mismatch."""
    rng = np.random.default_rng(seed)
    aa = rng.random() < 0.2
    sets = make_sets(rng, seed, aa, small)
    n_err = 0
    gap = [dict(gap_open1=0, gap_open2=0, gap_ext1=int(rng.integers(1, 5))), dict(gap_open1=int(rng.integers(2, 12)), gap_open2=0, gap_ext1=int(rng.integers(1, 4))),
           dict(), dict(gap_open1=int(rng.integers(3, 8)), gap_open2=int(rng.integers(12, 40)), gap_ext1=int(rng.integers(2, 4)), gap_ext2=1)][int(rng.integers(0, 4))]
    mode = int(rng.integers(0, 3))
    kw = dict(gap, aln_mode=mode)
    if mode != 1:
        r = rng.random()
        if r < 0.2:
            kw["extra_b"] = -1
        elif r < 0.5:
            kw.update(extra_b=int(rng.integers(2, 60)), extra_f=float(rng.choice([0.0, 0.01, 0.03])))
    if mode == 2 and rng.random() < 0.4:
        kw["zdrop"] = int(rng.integers(10, 100))
    if aa:
        kw.update(is_aa=True, score_matrix=workloads.BLOSUM62)
    amb = (not aa) and rng.random() < 0.3
    if amb:
        sets = [[(r[::-1].translate(COMP) if (j and rng.random() < 0.3) else r) for j, r in enumerate(s)] for s in sets]
    weights = [[rng.integers(1, 40, len(r)).astype(np.int32) for r in s] for s in sets] if rng.random() < 0.25 else None
    out_msa = bool(rng.random() < 0.7); out_cons = bool(rng.random() < 0.8) or not out_msa
    p = api.Params(**kw)
    dev = api.msa_batch(sets, p, out_cons=out_cons, out_msa=out_msa, n_threads=8, weights=weights, amb_strand=amb)
    nh = api.msa_timing()["n_host_sets"]; why = api.host_reasons() if nh else {}
    ref = api.msa_batch(sets, p, out_cons=out_cons, out_msa=out_msa, n_threads=8, weights=weights, amb_strand=amb, lib=shim)
    for i, (x, y) in enumerate(zip(dev, ref)):
        bad = None
        if x.status != 0 or y.status != 0:      # (both -5: the reference itself dies in its backtrack on such input -- z-drop breaks that leave the best cell's row behind)
            bad = f"status {x.status} / {y.status}" if x.status != y.status else None
            n_err += x.status == y.status
            if bad is None:
                continue
        elif out_cons and (x.cons_seq != y.cons_seq or x.cons_cov != y.cons_cov):
            bad = "consensus"
        elif out_msa and x.msa_seq != y.msa_seq:
            bad = "MSA rows"
        elif amb and list(x.is_rc) != list(y.is_rc):
            bad = "strand flags"
        if bad:
            raise AssertionError(f"MISMATCH ({bad}) iteration seed {seed} set {i}: {kw} amb={amb} weights={weights is not None} cons={out_cons} msa={out_msa} "
                                 f"shapes={[(len(s), max(map(len, s))) for s in sets]} host sets {nh}")
    return len(sets) - nh, nh, n_err, why


def main():
    ap = argparse.ArgumentParser(); ap.add_argument("--iters", type=int, default=100); ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--small", action="store_true", help="the shapes of the in-suite sweep (reads up to 600 bases, 20 per set)")
    a = ap.parse_args()
    lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))
    shim = H.cpu_shim_lib()
    n_dev = n_host = n_err = 0; hist = {}
    for it in range(a.iters):
        try:
            d, h, e, why = iteration(a.seed * 100000 + it, shim, a.small)
        except AssertionError as ex:
            print(ex, flush=True)
            sys.exit(1)
        n_dev += d; n_host += h; n_err += e
        if h:
            print(f"  iteration {it} (seed {a.seed * 100000 + it}): {h} set(s) through the host driver: {why}", flush=True)
        for k_, v_ in why.items():
            hist[k_] = hist.get(k_, 0) + v_
        if it % 20 == 19:
            print(f"{it + 1} iterations ok ({n_dev} sets on the device, {n_host} through the host driver{': ' + str(hist) if hist else ''})", flush=True)
    print(f"fuzz ok: {a.iters} iterations, {n_dev} sets on the device, {n_host} through the host driver, {n_err} sets on which both sides report the same error status")
    print(f"why sets left the device: {hist if hist else 'none did'}")


if __name__ == "__main__":
    main()
