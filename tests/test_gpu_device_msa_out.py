"""GPU (-m gpu): the device-resident read-set driver on the jobs it did not take before round 4 -- MSA output (read-id bitsets per edge,
MSA rank walk and row fill on the device), amino-acid alphabets (aligned groups of up to 26 nodes) and LOCAL alignment (the reference's own
Kahn row order rebuilt on the device before every read: local mode breaks score ties by row index).  Every result is compared with the
CPU build of the host layer whose aligner is the plain-C oracle (tests/cpu_shim.cpp): MSA rows byte for byte, consensus, coverage.
Reference code these follow: src/abpoa_graph.c:186-231 (row order), :315-375 (MSA rank), :418-484 (read ids), src/abpoa_output.c:103-166 (RC-MSA)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def engine():
    from abpoa_amd import ffi
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1
    ffi.check(lib.abpoa_hip_init(0))
    return lib


def _same(dev, ref, what, cons=True, msa=True):
    for i, (a, b) in enumerate(zip(dev, ref)):
        assert a.status == 0 and b.status == 0, f"{what}: set {i} status {a.status} / {b.status}"
        if msa:
            assert a.msa_len == b.msa_len, f"{what}: set {i}: {a.msa_len} MSA columns, oracle-backed run {b.msa_len}"
            assert a.msa_seq == b.msa_seq, f"{what}: MSA rows of set {i} differ"
        if cons:
            assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{what}: consensus of set {i} differs"


@pytest.mark.parametrize("lockstep", [0, 1], ids=["all_rounds_kernel", "lockstep_rounds"])
@pytest.mark.parametrize("out_cons", [False, True], ids=["msa_only", "msa_and_consensus"])
def test_global_msa_output_on_the_device(engine, monkeypatch, lockstep, out_cons):
    """Nucleotide reads, global banded alignment, MSA output (-r 1 / -r 2): ragged sets (3-40 reads: one and -- at 70 reads -- two words of read ids per
    edge), 2-18 % errors, affine and convex gaps.  The all-rounds kernel and the lock-step launches share the fuse body that keeps the read ids."""
    import helpers as H
    from abpoa_amd import api, synth
    monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", str(lockstep))
    shim = H.cpu_shim_lib()
    for kw, shapes in ((dict(gap_open1=4, gap_open2=0, gap_ext1=2), [(3 + (5 * i) % 38, 120 + 61 * i, 0.02 + 0.02 * (i % 8)) for i in range(12)] + [(70, 150, 0.08)]),
                       (dict(), [(6 + i, 300 + 90 * i, 0.12) for i in range(6)])):
        sets = [synth.make_read_set(31, i, n, ln, err) for i, (n, ln, err) in enumerate(shapes)]
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=out_cons, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, "device driver not used for every set"
        ref = api.msa_batch(sets, p, out_cons=out_cons, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"{kw} out_cons={out_cons}", cons=out_cons)


def test_protein_global_msa_and_consensus_on_the_device(engine):
    """27-code alphabet, BLOSUM62, global convex gaps: aligned groups of more than four nodes (substitution-heavy reads), consensus + MSA."""
    import helpers as H
    from abpoa_amd import api, synth, workloads
    shim = H.cpu_shim_lib()
    p = api.Params(is_aa=True, score_matrix=workloads.BLOSUM62)
    sets = [synth.make_read_set(37, i, 12 + i % 9, 150 + 40 * i, alphabet=synth.AA, rates=(0.10 + 0.02 * (i % 4), 0.02, 0.02)) for i in range(10)]
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0, "device driver not used for every set"
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
    _same(dev, ref, "protein global")


@pytest.mark.parametrize("name,kw,mk", [
    ("aa_blosum_convex", dict(aln_mode=1, is_aa=True, score_matrix="BLOSUM62"), lambda i: dict(n_reads=8 + i % 23, length=60 + 37 * i, alphabet="AA", rates=(0.05 + 0.01 * (i % 5), 0.03, 0.03))),
    ("aa_blosum_affine", dict(aln_mode=1, is_aa=True, score_matrix="BLOSUM62", gap_open1=11, gap_open2=0, gap_ext1=1), lambda i: dict(n_reads=10 + i % 7, length=100 + 38 * i, alphabet="AA", rates=(0.08, 0.02, 0.04))),
    ("nt_local_convex", dict(aln_mode=1), lambda i: dict(n_reads=6 + i % 11, length=90 + 43 * i, err=0.04 + 0.02 * (i % 6))),
    ("nt_local_affine", dict(aln_mode=1, gap_open1=4, gap_open2=0, gap_ext1=2), lambda i: dict(n_reads=12, length=200 + 28 * i, err=0.10)),
])
def test_local_mode_msa_on_the_device(engine, name, kw, mk):
    """Local alignment (-m 1: no band), MSA output, ragged read-sets of 60-540 residues: the shape of BASELINE.json configs[4] and around it.  The row
    order is the reference's Kahn walk rebuilt on the device before every read (poa_order_kernel)."""
    import helpers as H
    from abpoa_amd import api, synth, workloads
    shim = H.cpu_shim_lib()
    kw = dict(kw)
    if kw.get("score_matrix") == "BLOSUM62":
        kw["score_matrix"] = workloads.BLOSUM62
    p = api.Params(**kw)
    sets = []
    for i in range(12):
        a = mk(i)
        if a.get("alphabet") == "AA":
            a["alphabet"] = synth.AA
        sets.append(synth.make_read_set(43, i, **a))
    for out_cons in (False, True):
        dev = api.msa_batch(sets, p, out_cons=out_cons, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, f"{name}: device driver not used for every set"
        ref = api.msa_batch(sets, p, out_cons=out_cons, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"{name} out_cons={out_cons}", cons=out_cons)
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4)      # consensus only, local mode
    assert api.msa_timing()["n_host_sets"] == 0
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4, lib=shim)
    _same(dev, ref, f"{name} consensus only", msa=False)


def test_config5_sample_on_the_device_against_reference_digests(engine):
    """BASELINE.json configs[4]: the first 48 sets (30 x 500 aa, local convex BLOSUM62, MSA output) through the device driver against the committed
    digests of the reference's own output for those sets."""
    from abpoa_amd import api, synth, workloads
    wl = workloads.WORKLOADS["cfg5"]
    dig = workloads.load_digests("cfg5")
    assert dig is not None
    sets = [synth.make_read_set(1, i, **synth.CONFIGS[5]) for i in range(48)]
    p = api.Params(**wl["params"])
    res = api.msa_batch(sets, p, out_cons=False, out_msa=True, n_threads=8)
    assert api.msa_timing()["n_host_sets"] == 0, "device driver not used for every set"
    for i, r in enumerate(res):
        assert r.status == 0
        assert workloads.output_sha(api.format_output(r, [f"r{j}" for j in range(len(sets[i]))], False, True)) == dig[i], f"set {i}: output differs from the reference's"


def test_device_row_order_graph_and_msa_checks_after_every_read(engine):
    """ABPOA_HIP_DEVSYNC=1 (child process: the mode is read from the environment and reports on stderr): after every read the library compares the row
    order the order kernel wrote with the host graph's Kahn walk, the device graph with the host graph fed the same cigars, and at the end the device
    MSA with the host routine."""
    code = ("import os,sys; sys.path.insert(0, %r)\n"
            "from abpoa_amd import api, ffi, synth, workloads\n"
            "lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))\n"
            "sets = [synth.make_read_set(3, i, 9, 120 + 30 * i, alphabet=synth.AA, rates=(0.08, 0.03, 0.03)) for i in range(5)]\n"
            "p = api.Params(aln_mode=1, is_aa=True, score_matrix=workloads.BLOSUM62)\n"
            "r = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)\n"
            "print('OK', all(x.status == 0 for x in r), api.msa_timing()['n_host_sets'])\n" % ROOT)
    env = dict(os.environ, ABPOA_HIP_DEVSYNC="1", ABPOA_HIP_HOSTGRAPH="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-3000:]
    assert "OK True 0" in p.stdout, (p.stdout, p.stderr[-3000:])
    for what in ("row order check ok", "graph check ok", "msa check ok", "consensus check ok"):
        assert what in p.stderr, (what, p.stderr[-3000:])
    assert "FAILED" not in p.stderr, p.stderr[-3000:]


def test_strict_mode_reports_instead_of_slowing_down(engine, monkeypatch):
    """ABPOA_HIP_STRICT=1: a job whose options are the host driver's (here: linear gaps without a band) fails with ABPOA_HIP_ESTRICT instead of running at host-driver
    speed unnoticed; without the switch it runs and reports every set in abpoa_hip_msa_timing_t.n_host_sets.  A device-driver job is unaffected."""
    from abpoa_amd import api, ffi, synth
    sets = [synth.make_read_set(53, i, 6, 200, 0.05) for i in range(4)]
    monkeypatch.setenv("ABPOA_HIP_NO_DEVICE_GENERAL", "1")      # (linear gaps without a band: the general kernel's job, which this switch keeps on the host driver)
    lin = api.Params(gap_open1=0, gap_open2=0, gap_ext1=2, extra_b=-1)
    r = api.msa_batch(sets, lin, n_threads=4)
    assert all(x.status == 0 for x in r) and api.msa_timing()["n_host_sets"] == len(sets)
    monkeypatch.setenv("ABPOA_HIP_STRICT", "1")
    with pytest.raises(ffi.EngineError) as ei:
        api.msa_batch(sets, lin, n_threads=4)
    assert "(-6)" in str(ei.value) and "ABPOA_HIP_STRICT" in str(ei.value)
    ok = api.msa_batch(sets, api.Params(gap_open1=4, gap_open2=0, gap_ext1=2), n_threads=4)
    assert all(x.status == 0 for x in ok) and api.msa_timing()["n_host_sets"] == 0


@pytest.mark.parametrize("lockstep", [0, 1], ids=["all_rounds_kernel", "lockstep_rounds"])
def test_per_base_weights_on_the_device(engine, monkeypatch, lockstep):
    """The reference's -Q (base qualities as edge weights, src/abpoa_align.c:462-467; abpoa_graph.c:486-499, :634-667): the device fuse phase adds the weight of
    the base an edge leads to.  Ragged sets, some reads (and one whole set) without weights, consensus + coverage + MSA against the oracle-backed run; and the
    reference CLI's own -Q outputs (goldens out_qv_*) through the device driver."""
    import numpy as np
    import helpers as H
    from abpoa_amd import api, seqio, synth
    monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", str(lockstep))
    shim = H.cpu_shim_lib()
    rng = np.random.default_rng(7)
    sets = [synth.make_read_set(59, i, 5 + i % 9, 180 + 50 * i, 0.04 + 0.02 * (i % 5)) for i in range(10)]
    weights = [[rng.integers(1, 41, len(r)).astype(np.int32) for r in s] for s in sets]
    weights[3] = None
    for kw in (dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict()):
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=weights)
        assert api.msa_timing()["n_host_sets"] == 0, "device driver not used for every set"
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=weights, lib=shim)
        _same(dev, ref, f"weights {kw}")
    D = H.GOLDEN_DIR
    for name, out_msa in (("out_qv_cons", False), ("out_qv_msa", True)):
        names, seqs, quals = seqio.read_fastx(os.path.join(D, name, "input.fq"))
        w = [[seqio.qv_weights(s, q) for s, q in zip(seqs, quals)]]
        r = api.msa_batch([seqs], api.Params(gap_open1=4, gap_open2=0, gap_ext1=2), out_cons=True, out_msa=out_msa, weights=w)
        assert api.msa_timing()["n_host_sets"] == 0
        assert api.format_output(r[0], names, True, out_msa) == open(os.path.join(D, name, "output.txt")).read(), name


def test_two_contexts_on_two_host_threads(engine):
    """abpoa_hip_ctx_t: two host threads, a context each (own device queue, timing and last error), different jobs at the same time -- one narrow-band
    job that takes the all-rounds kernel (per-queue __constant__ argument records) and one local amino-acid MSA job -- three calls each; results equal
    those of the process-wide entry, each context's timing describes its own job, and a failing call leaves its message in its own context only."""
    import threading
    from abpoa_amd import api, ffi, synth, workloads
    jobs = [(api.Params(gap_open1=4, gap_open2=0, gap_ext1=2), [synth.make_read_set(61, i, 12, 600, 0.06) for i in range(24)], dict(out_cons=True, out_msa=False)),
            (api.Params(aln_mode=1, is_aa=True, score_matrix=workloads.BLOSUM62), [synth.make_read_set(67, i, 10, 200, alphabet=synth.AA, rates=(0.06, 0.03, 0.03)) for i in range(16)], dict(out_cons=False, out_msa=True))]
    want = [api.msa_batch(sets, p, n_threads=4, **kw) for p, sets, kw in jobs]
    ctxs = [api.BatchContext() for _ in jobs]
    got, errs = [None, None], []

    def work(i):
        try:
            p, sets, kw = jobs[i]
            for _ in range(3):
                got[i] = ctxs[i].msa_batch(sets, p, n_threads=4, **kw)
        except Exception as ex:      # noqa: BLE001
            errs.append(repr(ex))

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t.start() for t in th]; [t.join() for t in th]
    assert not errs, errs
    for i, (p, sets, kw) in enumerate(jobs):
        for a, b in zip(got[i], want[i]):
            assert a.status == 0 and a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov and a.msa_seq == b.msa_seq
        tm = ctxs[i].timing()
        assert tm["n_host_sets"] == 0 and tm["n_rounds"] == max(len(s) for s in sets) - 1, tm
    with pytest.raises(ffi.EngineError):      # a malformed job: the message stays with the context that made the call
        ctxs[0].msa_batch([[""]], jobs[0][0])
    assert ctxs[0].lib.abpoa_hip_ctx_last_error(ctxs[0].h) != b"" and ctxs[1].lib.abpoa_hip_ctx_last_error(ctxs[1].h) == b""
    [c.close() for c in ctxs]


@pytest.mark.parametrize("env", [{"ABPOA_HIP_ORDER_LDS": "0"}, {"ABPOA_HIP_ORDER_LDS": "0", "ABPOA_HIP_ORDER_CAP": "0"}, {"ABPOA_HIP_ORDER_CAP": "400"}],
                         ids=["general_walk_lds_tables", "general_walk_global_tables", "mixed_by_graph_size"])
def test_order_and_rank_walks_in_every_form(engine, monkeypatch, env):
    """The row-order kernel has three forms -- everything in LDS with one counter per aligned group (the default for graphs that fit), the general walk with
    its tables in LDS, and the general walk with its tables in memory (graphs beyond the LDS capacity) -- and the MSA rank walk two.  The switches force
    each; ABPOA_HIP_ORDER_CAP=400 makes the form change inside one job as the graphs grow.  Local amino-acid and nucleotide MSAs against the oracle-backed run."""
    import helpers as H
    from abpoa_amd import api, synth, workloads
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    shim = H.cpu_shim_lib()
    jobs = [(api.Params(aln_mode=1, is_aa=True, score_matrix=workloads.BLOSUM62), [synth.make_read_set(73, i, 9 + i % 8, 120 + 45 * i, alphabet=synth.AA, rates=(0.08, 0.03, 0.03)) for i in range(8)]),
            (api.Params(aln_mode=1, gap_open1=4, gap_open2=0, gap_ext1=2), [synth.make_read_set(79, i, 8 + i % 5, 150 + 50 * i, 0.08) for i in range(8)])]
    for p, sets in jobs:
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"order walk {env}")
