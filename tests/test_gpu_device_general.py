"""GPU (-m gpu): the device-resident read-set driver on the jobs that are outside the fast row loops -- linear gaps (reference simd_abpoa_lg_dp,
src/simd_abpoa_align.c:701-779), extension mode with and without z-drop (:1018-1026), global alignment without a band (-b -1), local alignment of reads
longer than the local row loop holds.  They run in the general kernel (rows_general.h), one launch per round, with the graph resident on the device like
every other job: the prepare phase also writes the successor lists that kernel hands its band state through, and extension mode gets the reference's own
row order (its best cell is the first row that reaches the maximum).  Every result is compared with the CPU build of the host layer whose aligner is the
plain-C oracle (tests/cpu_shim.cpp): MSA rows byte for byte, consensus, coverage -- and `n_host_sets == 0`: nothing went through the host driver."""
import pytest

pytestmark = pytest.mark.gpu

EXTEND = 2


@pytest.fixture(scope="module")
def engine():
    from abpoa_amd import ffi
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1
    ffi.check(lib.abpoa_hip_init(0))
    return lib


def _same(dev, ref, what):
    for i, (a, b) in enumerate(zip(dev, ref)):
        assert a.status == 0 and b.status == 0, f"{what}: set {i} status {a.status} / {b.status}"
        assert a.msa_len == b.msa_len, f"{what}: set {i}: {a.msa_len} MSA columns, oracle-backed run {b.msa_len}"
        assert a.msa_seq == b.msa_seq, f"{what}: MSA rows of set {i} differ"
        assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{what}: consensus of set {i} differs"


def _nt_sets(seed, n=10, length=lambda i: 150 + 70 * i):
    from abpoa_amd import synth
    return [synth.make_read_set(seed, i, 4 + (3 * i) % 13, length(i), 0.03 + 0.025 * (i % 6)) for i in range(n)]


VARIANTS = [
    ("linear_global_banded", dict(gap_open1=0, gap_open2=0, gap_ext1=2), None),
    ("linear_global_unbanded", dict(gap_open1=0, gap_open2=0, gap_ext1=3, extra_b=-1), None),
    ("linear_local", dict(aln_mode=1, gap_open1=0, gap_open2=0, gap_ext1=2), None),
    ("affine_global_unbanded", dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=-1), None),
    ("convex_global_unbanded", dict(extra_b=-1), None),
    ("affine_extend", dict(aln_mode=EXTEND, gap_open1=4, gap_open2=0, gap_ext1=2), None),
    ("convex_extend", dict(aln_mode=EXTEND), None),
    ("convex_extend_zdrop", dict(aln_mode=EXTEND, zdrop=40), None),
    ("linear_extend_unbanded", dict(aln_mode=EXTEND, gap_open1=0, gap_open2=0, gap_ext1=2, extra_b=-1), None),
    ("convex_local_long_reads", dict(aln_mode=1), lambda i: 640 + 45 * i),
    ("affine_local_long_reads", dict(aln_mode=1, gap_open1=6, gap_open2=0, gap_ext1=2), lambda i: 600 + 60 * i),
]


@pytest.mark.parametrize("name,kw,length", VARIANTS, ids=[v[0] for v in VARIANTS])
def test_general_kernel_jobs_stay_on_the_device(engine, name, kw, length):
    import helpers as H
    from abpoa_amd import api
    shim = H.cpu_shim_lib()
    sets = _nt_sets(71, 8, length) if length else _nt_sets(71)
    p = api.Params(**kw)
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0, f"{name}: not every set ran on the device-resident driver"
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
    _same(dev, ref, name)


def test_protein_linear_and_extend_on_the_device(engine):
    """27-code alphabet (aligned groups of up to 26 nodes), BLOSUM62: linear gaps global, convex extension."""
    import helpers as H
    from abpoa_amd import api, synth, workloads
    shim = H.cpu_shim_lib()
    sets = [synth.make_read_set(73, i, 6 + i % 9, 90 + 40 * i, alphabet=synth.AA, rates=(0.08 + 0.02 * (i % 4), 0.02, 0.03)) for i in range(8)]
    for kw in (dict(gap_open1=0, gap_open2=0, gap_ext1=4), dict(aln_mode=EXTEND)):
        p = api.Params(is_aa=True, score_matrix=workloads.BLOSUM62, **kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"protein {kw}")


def test_general_kernel_equals_the_fast_row_loops_on_their_jobs(engine, monkeypatch):
    """ABPOA_HIP_DEVICE_GENERAL=1 sends banded global jobs (the fast row loops' own) through the general kernel on the device: same consensus and MSA, and
    both equal the oracle-backed run.  (Their cigars: tests/test_gpu_device_msa.py::test_device_driver_cigars_equal_the_oracle_backed_run.)"""
    import helpers as H
    from abpoa_amd import api
    shim = H.cpu_shim_lib()
    sets = _nt_sets(79, 8)
    for kw in (dict(), dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict(gap_open1=0, gap_open2=0, gap_ext1=2)):
        p = api.Params(**kw)
        monkeypatch.setenv("ABPOA_HIP_DEVICE_GENERAL", "0")
        fast = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        monkeypatch.setenv("ABPOA_HIP_DEVICE_GENERAL", "1")
        gen = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(gen, ref, f"general kernel {kw}")
        _same(fast, ref, f"fast loops {kw}")


LINEAR_FAST = [
    ("e2_1kb_5pct", dict(gap_ext1=2), 12, 1000, 0.05),
    ("e2_1kb_15pct", dict(gap_ext1=2), 10, 900, 0.15),          # many rows with three and more predecessors, bands that open beyond 64 columns
    ("e1_600_10pct", dict(gap_ext1=1), 16, 600, 0.10),          # the smallest extension the fast loops take (the row arg-max is read before the in-row scan)
    ("e5_2500_8pct", dict(gap_ext1=5, mismatch=7), 6, 2500, 0.08),      # w = 35: the widest band of the narrow loop
    ("int32_scores_800", dict(gap_ext1=3, match=45, mismatch=60), 8, 800, 0.07),      # 800 x 45 leaves int16
    ("e0_general_kernel", dict(gap_ext1=0), 6, 300, 0.05),      # extension 0: ties between a cell and its left neighbour -- stays with the general kernel
    ("extend_e2_700", dict(gap_ext1=2, aln_mode=EXTEND), 9, 700, 0.08),      # extension mode: the global rows + the running best cell (rows_fast.h commit_row)
    ("extend_zdrop_e3_500", dict(gap_ext1=3, aln_mode=EXTEND, zdrop=30), 9, 500, 0.12),
]


@pytest.mark.parametrize("name,kw,n_reads,length,err", LINEAR_FAST, ids=[v[0] for v in LINEAR_FAST])
def test_linear_gaps_on_the_fast_row_loops(engine, monkeypatch, name, kw, n_reads, length, err):
    """Banded global alignment with linear gaps (reference simd_abpoa_lg_dp, src/simd_abpoa_align.c:701-779, backtrack :109-190) on the narrow row loop
    (rows_fast.h GAP = 0: one 64-lane prefix-max scan on H per row, H-only records, lane-parallel linear backtrack steps): consensus and MSA rows equal the
    oracle-backed run AND the general kernel's (ABPOA_HIP_DEVICE_GENERAL=1) on 5 - 15 % reads, int16 and int32 scores, extension penalties 1 - 5."""
    import helpers as H
    from abpoa_amd import api, synth
    shim = H.cpu_shim_lib()
    sets = [synth.make_read_set(97, i, n_reads + i, length + 13 * i, err) for i in range(4)]
    p = api.Params(gap_open1=0, gap_open2=0, **kw)
    monkeypatch.setenv("ABPOA_HIP_DEVICE_GENERAL", "0")
    fast = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0
    monkeypatch.setenv("ABPOA_HIP_DEVICE_GENERAL", "1")
    gen = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
    _same(gen, ref, f"linear {name}: general kernel")
    _same(fast, ref, f"linear {name}: fast row loops")


def test_general_jobs_can_be_kept_off_the_device(engine, monkeypatch):
    """ABPOA_HIP_NO_DEVICE_GENERAL=1: the jobs of the general kernel go through the host driver as before round 4 (and are counted)."""
    from abpoa_amd import api
    sets = _nt_sets(83, 4)
    monkeypatch.setenv("ABPOA_HIP_NO_DEVICE_GENERAL", "1")
    r = api.msa_batch(sets, api.Params(gap_open1=0, gap_open2=0, gap_ext1=2, extra_b=-1), n_threads=4)      # (linear gaps without a band: banded linear jobs take the fast row loops since round 5)
    assert all(x.status == 0 for x in r) and api.msa_timing()["n_host_sets"] == len(sets)


# ---------------------------------------------------------------------------------------------------------------- -s (ambiguous strand)
def _flip(read):
    return read[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def _mixed_strand_sets(seed, shapes):
    """Read-sets in which about a third of the reads (never the first) are given as their reverse complement."""
    import numpy as np
    from abpoa_amd import synth
    rng = np.random.default_rng(seed)
    sets = []
    for i, (n, ln, err) in enumerate(shapes):
        reads = list(synth.make_read_set(seed, i, n, ln, err))
        for j in range(1, n):
            if rng.random() < 0.35:
                reads[j] = _flip(reads[j])
        sets.append(reads)
    return sets


def test_strand_retry_goldens_on_the_device(engine):
    """The reference CLI's own -s outputs (goldens out_rc_cons / out_rc_msa / out_rc_long_msa: the last one is long noisy reads whose retry inherits the
    forward run's band state) through the device-resident driver: forward alignment in the fast row loop, strand check, retry in the general kernel on the
    reverse complement, the better strand fused (reference src/abpoa_align.c:315-336)."""
    import os
    import helpers as H
    from abpoa_amd import api, seqio
    D = H.GOLDEN_DIR
    for name, src, out_msa, want_rc in (("out_rc_cons", "out_rc_cons", False, [2, 5, 8]), ("out_rc_msa", "out_rc_cons", True, [2, 5, 8]), ("out_rc_long_msa", "out_rc_long_msa", True, [2, 5])):
        names, seqs, _ = seqio.read_fastx(os.path.join(D, src, "input.fa"))
        r = api.msa_batch([seqs], api.Params(), out_cons=True, out_msa=out_msa, amb_strand=True)[0]
        assert api.msa_timing()["n_host_sets"] == 0, f"{name}: not on the device-resident driver"
        assert [i for i, f in enumerate(r.is_rc) if f] == want_rc, name
        assert api.format_output(r, names, True, out_msa) == open(os.path.join(D, name, "output.txt")).read(), name


@pytest.mark.parametrize("name,kw", [("convex_banded", dict()), ("affine_banded", dict(gap_open1=4, gap_open2=0, gap_ext1=2)), ("linear_banded", dict(gap_open1=0, gap_open2=0, gap_ext1=2)),
                                     ("affine_unbanded", dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=-1)), ("convex_local", dict(aln_mode=1)), ("convex_extend", dict(aln_mode=EXTEND))])
def test_mixed_strand_sets_on_the_device(engine, name, kw):
    """-s on ragged read-sets with about a third of the reads reverse-complemented, every driver form the option meets: forward run in the narrow / wide fast
    row loops (band state handed to the retry), in the local row loop (no band), in the general kernel (linear gaps, no band, extension); with and without
    per-base weights (the retry reverses them, reference :323-326).  MSA rows, consensus, coverage and the strand flags against the oracle-backed run."""
    import numpy as np
    import helpers as H
    from abpoa_amd import api
    shim = H.cpu_shim_lib()
    shapes = [(5 + (3 * i) % 11, 140 + 80 * i, 0.03 + 0.02 * (i % 5)) for i in range(9)] + [(6, 2600, 0.10), (5, 4200, 0.12)]      # (the last two: band half-widths of 40+, the wide row loop)
    if kw.get("aln_mode") == 1:
        shapes = shapes[:9]
    sets = _mixed_strand_sets(89, shapes)
    rng = np.random.default_rng(3)
    weights = [[rng.integers(1, 41, len(r)).astype(np.int32) for r in s] for s in sets]
    p = api.Params(**kw)
    for w in (None, weights):
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=w, amb_strand=True)
        assert api.msa_timing()["n_host_sets"] == 0, f"{name}: not every set ran on the device-resident driver"
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=w, amb_strand=True, lib=shim)
        _same(dev, ref, f"{name} weights={'yes' if w else 'no'}")
        assert [list(a.is_rc) for a in dev] == [list(b.is_rc) for b in ref], f"{name}: strand flags differ"
        assert sum(sum(a.is_rc) for a in dev) > 5, "the inputs were meant to have reverse-complemented reads"


def test_strand_retry_can_be_kept_off_the_device(engine, monkeypatch):
    from abpoa_amd import api
    sets = _mixed_strand_sets(97, [(6, 300, 0.05)] * 3)
    monkeypatch.setenv("ABPOA_HIP_NO_DEVICE_STRAND", "1")
    r = api.msa_batch(sets, api.Params(), n_threads=4, amb_strand=True)
    assert all(x.status == 0 for x in r) and api.msa_timing()["n_host_sets"] == len(sets)


def test_random_option_mixes_and_degenerate_inputs(engine):
    """A seeded sweep over the option space (gap model x alignment mode x band on / off x -s x weights x output kind) on small ragged sets that include the
    degenerate shapes -- one read, two reads, reads of 1-5 bases, identical reads, reads of N only, a read far longer than the rest -- device-resident driver
    against the oracle-backed run.  Whatever the device cannot hold may go to the host driver (counted, not asserted here); results must be equal either way."""
    import numpy as np
    import helpers as H
    from abpoa_amd import api, synth, workloads
    shim = H.cpu_shim_lib()
    rng = np.random.default_rng(2024)

    def rnd_read(n):
        return "".join("ACGT"[c] for c in rng.integers(0, 4, n))
    base_sets = [
        [rnd_read(40)],                                              # one read
        [rnd_read(30), rnd_read(33)],                                # two unrelated reads
        ["A", "C", "A", "AC"],                                       # reads of one and two bases
        ["ACGTA", "ACGA", "ACGTTA", "CGTA", "ACGTA"],
        ["NNNNNNNN", "NNNNNNN", "NNNNNNNNN"],
        [rnd_read(60)] * 5,                                          # identical reads
        list(synth.make_read_set(101, 0, 7, 90, 0.08)) + [rnd_read(400)],      # one read far longer than the rest (and unrelated)
        list(synth.make_read_set(101, 1, 9, 260, 0.12)),
        list(synth.make_read_set(101, 2, 5, 700, 0.05)),
    ]
    n_dev = n_host = 0
    aa_sets = [list(synth.make_read_set(103, i, 3 + 2 * i, 30 + 45 * i, alphabet=synth.AA, rates=(0.08, 0.03, 0.03))) for i in range(5)] + [["MKV", "MKLV", "M"]]
    for it in range(40):
        gap = [dict(gap_open1=0, gap_open2=0, gap_ext1=int(rng.integers(1, 5))), dict(gap_open1=int(rng.integers(2, 9)), gap_open2=0, gap_ext1=int(rng.integers(1, 4))), dict()][it % 3]
        mode = int(rng.integers(0, 3))
        kw = dict(gap, aln_mode=mode)
        if mode != 1 and rng.random() < 0.4:
            kw["extra_b"] = -1
        if mode == EXTEND and rng.random() < 0.5:
            kw["zdrop"] = int(rng.integers(10, 80))
        amb = bool(rng.random() < 0.4)
        out_msa = bool(rng.random() < 0.7)
        aa = it % 4 == 3
        if aa:
            amb = False
            kw.update(is_aa=True, score_matrix=workloads.BLOSUM62)
        sets = [list(s) for s in (aa_sets if aa else base_sets)]
        if amb:
            sets = [[(_flip(r) if (j and rng.random() < 0.3) else r) for j, r in enumerate(s)] for s in sets]
        weights = [[rng.integers(1, 30, len(r)).astype(np.int32) for r in s] for s in sets] if rng.random() < 0.4 else None
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=out_msa, n_threads=4, weights=weights, amb_strand=amb)
        nh = api.msa_timing()["n_host_sets"]
        n_host += nh; n_dev += len(sets) - nh
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=out_msa, n_threads=4, weights=weights, amb_strand=amb, lib=shim)
        what = f"iteration {it}: {kw} amb={amb} msa={out_msa} weights={weights is not None}"
        for i, (a, b) in enumerate(zip(dev, ref)):
            assert a.status == 0 and b.status == 0, f"{what}: set {i} status {a.status} / {b.status}"
            assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{what}: consensus of set {i} differs"
            if out_msa:
                assert a.msa_seq == b.msa_seq, f"{what}: MSA rows of set {i} differ"
            if amb:
                assert list(a.is_rc) == list(b.is_rc), f"{what}: strand flags of set {i} differ"
    assert n_dev > 8 * n_host, (n_dev, n_host)      # (the sweep is meant to exercise the device-resident driver)


# ---------------------------------------------------------------------------------------------------------------- reads with ragged ends
def _ragged_sets(seed, shapes, frac):
    """every read but the first is a random substring of its noisy full-length version: up to `frac` of the length cut from each end"""
    import numpy as np
    from abpoa_amd import synth
    rng = np.random.default_rng(seed)
    sets = []
    for i, (n, ln, err) in enumerate(shapes):
        reads = list(synth.make_read_set(seed, i, n, ln, err))
        out = [reads[0]]
        for r in reads[1:]:
            a = int(rng.integers(0, int(frac * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(frac * len(r)) + 1))
            out.append(r[a:b])
        sets.append(out)
    return sets


@pytest.mark.parametrize("lockstep", [0, 1], ids=["all_rounds_kernel", "lockstep_rounds"])
def test_reads_with_ragged_ends_stay_on_the_device(engine, monkeypatch, lockstep):
    """Reads that do not start and end on the same node (window-cut or partly sequenced reads): the source collects an out-edge, the sink an in-edge per
    distinct end -- far more than the 16 / 15 slots of a node (PoaSet.term0: the terminal pools) -- and the band, anchored at `qlen - remaining length`, sits
    as far from the path as the lengths differ (PoaSet.band_extra: wider arena rows, the wide row loop).  60 reads per set so that the source passes 16 edges;
    consensus, MSA and -- with 70 reads -- the rank walk over more than 64 out-edges of one node, against the oracle-backed run."""
    import helpers as H
    from abpoa_amd import api
    monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", str(lockstep))
    shim = H.cpu_shim_lib()
    # (the last two: rows a few hundred columns wider than 2 w -- beyond the wide row loop's 448 columns they take the narrow kernel's chunk-by-chunk bodies)
    sets = _ragged_sets(107, [(60, 300, 0.04), (40, 500, 0.06), (70, 240, 0.03), (25, 900, 0.05), (12, 1500, 0.08), (24, 3000, 0.05), (10, 5000, 0.08)], 0.12)
    sets.append([sets[2][0]] + [sets[2][0][k:] for k in range(1, 80)])      # 80 reads, every one starting one base later: 80 out-edges of the source
    for kw in (dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict()):
        p = api.Params(**kw)
        for out_msa in (False, True):
            dev = api.msa_batch(sets, p, out_cons=True, out_msa=out_msa, n_threads=4)
            assert api.msa_timing()["n_host_sets"] == 0, f"{kw}: not every set ran on the device-resident driver"
            ref = api.msa_batch(sets, p, out_cons=True, out_msa=out_msa, n_threads=4, lib=shim)
            for i, (a, b) in enumerate(zip(dev, ref)):
                assert a.status == 0 and b.status == 0
                assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{kw}: consensus of set {i} differs"
                if out_msa:
                    assert a.msa_seq == b.msa_seq, f"{kw}: MSA rows of set {i} differ"


def test_ragged_local_and_general_jobs(engine):
    """The terminal pools under the other drivers of the device path: local mode (order walk over a source with dozens of out-edges: the time stamps of its
    edges must not collide with later queue positions), linear gaps and extension mode (general kernel: successor lists of row 0)."""
    import helpers as H
    from abpoa_amd import api
    shim = H.cpu_shim_lib()
    sets = _ragged_sets(109, [(45, 260, 0.04), (30, 420, 0.06), (50, 180, 0.03)], 0.15)
    for kw in (dict(aln_mode=1), dict(aln_mode=1, gap_open1=5, gap_open2=0, gap_ext1=2), dict(gap_open1=0, gap_open2=0, gap_ext1=2), dict(aln_mode=EXTEND)):
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, f"{kw}: not every set ran on the device-resident driver"
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"ragged {kw}")


def test_random_option_mixes_on_ragged_reads(engine):
    """The seeded option sweep again, on reads with ragged ends (5-25 % cut per end, 8-70 reads, 60-1200 bases): every gap model, alignment mode, band on / off,
    -s, weights; consensus, MSA and strand flags against the oracle-backed run."""
    import numpy as np
    import helpers as H
    from abpoa_amd import api
    shim = H.cpu_shim_lib()
    rng = np.random.default_rng(4242)
    n_dev = n_host = 0
    for it in range(24):
        shapes = [(int(rng.integers(8, 70)), int(rng.integers(60, 1200)), float(rng.uniform(0.02, 0.12))) for _ in range(5)]
        sets = _ragged_sets(1000 + it, shapes, float(rng.uniform(0.05, 0.25)))
        gap = [dict(gap_open1=0, gap_open2=0, gap_ext1=int(rng.integers(1, 5))), dict(gap_open1=int(rng.integers(2, 9)), gap_open2=0, gap_ext1=int(rng.integers(1, 4))), dict()][it % 3]
        mode = int(rng.integers(0, 3))
        kw = dict(gap, aln_mode=mode)
        if mode != 1 and rng.random() < 0.3:
            kw["extra_b"] = -1
        if mode == EXTEND and rng.random() < 0.5:
            kw["zdrop"] = int(rng.integers(10, 80))
        amb = bool(rng.random() < 0.3)
        if amb:
            sets = [[(_flip(r) if (j and rng.random() < 0.3) else r) for j, r in enumerate(s)] for s in sets]
        weights = [[rng.integers(1, 30, len(r)).astype(np.int32) for r in s] for s in sets] if rng.random() < 0.3 else None
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=weights, amb_strand=amb)
        nh = api.msa_timing()["n_host_sets"]
        n_host += nh; n_dev += len(sets) - nh
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, weights=weights, amb_strand=amb, lib=shim)
        what = f"iteration {it}: {kw} amb={amb} weights={weights is not None} shapes={shapes}"
        for i, (a, b) in enumerate(zip(dev, ref)):
            assert a.status == 0 and b.status == 0, f"{what}: set {i} status {a.status} / {b.status}"
            assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{what}: consensus of set {i} differs"
            assert a.msa_seq == b.msa_seq, f"{what}: MSA rows of set {i} differ"
            if amb:
                assert list(a.is_rc) == list(b.is_rc), f"{what}: strand flags of set {i} differ"
    assert n_dev > 4 * n_host, (n_dev, n_host)


def test_more_start_nodes_than_the_terminal_pools_hold(engine):
    """300 reads that all start on different nodes: the source would need 300 out-edges, the pools stop at 250 (the counts are bytes) -- the set goes to the last
    pass of the ladder, overflows there too and is handed to the host driver; the result is the reference's all the same; its neighbour in the job stays on the device."""
    import helpers as H
    from abpoa_amd import api, synth
    shim = H.cpu_shim_lib()
    base = synth.make_read_set(113, 0, 1, 420, 0.0)[0]
    many = [base] + [base[k:] for k in range(1, 300)]
    sets = [many, list(synth.make_read_set(113, 1, 12, 300, 0.05))]
    p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 1
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4, lib=shim)
    for a, b in zip(dev, ref):
        assert a.status == 0 and a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov


def test_inner_nodes_with_more_edges_than_the_edge_slots(engine):
    """25 reads that each delete a different number of bases behind (in front of) the same node: that node collects 26 out-edges (in-edges), more than the 16 (15)
    slots a node has in the regular passes.  Such sets used to go to the host driver; now the pass that finds the full list sends them straight to the last
    pass of the ladder, whose layout has a slot per read at every node (msa_device.cpp `roomy`; ABPOA_HIP_VERBOSE shows `pass 1 ... 2 sets outgrew` then
    `pass 4 ... 0 sets`), and nothing leaves the device.  Global affine / convex, local, extension; consensus and MSA; against the oracle-backed run."""
    import helpers as H
    from abpoa_amd import api, synth
    shim = H.cpu_shim_lib()
    base = synth.make_read_set(211, 0, 1, 420, 0.0)[0]
    fan_out = [base] + [base[:200] + base[200 + k:] for k in range(1, 26)]
    fan_in = [base] + [base[:200 - k] + base[200:] for k in range(1, 26)]
    sets = [fan_out, fan_in, fan_out + fan_in[1:], list(synth.make_read_set(211, 1, 12, 300, 0.05))]
    for kw in (dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict(), dict(aln_mode=1), dict(aln_mode=2)):
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, (kw, api.host_reasons())
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"fan of edges {kw}")


def test_hundreds_of_reads_per_set(engine):
    """400 and 130 reads per set (seven and three words of read ids per edge, read counts beyond a byte), MSA + consensus, uncut and ragged."""
    import helpers as H
    from abpoa_amd import api, synth
    shim = H.cpu_shim_lib()
    sets = [list(synth.make_read_set(127, 0, 400, 180, 0.04)), list(synth.make_read_set(127, 1, 130, 420, 0.07))] + _ragged_sets(131, [(200, 260, 0.05)], 0.1)
    for kw in (dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict(aln_mode=1)):
        p = api.Params(**kw)
        dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, kw
        ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
        _same(dev, ref, f"many reads {kw}")


def test_reads_beyond_the_fast_loops_query_limit(engine):
    """Reads of 34 000 bases (the fast row loops keep query codes in LDS up to 32 000): the job runs device-resident in the general kernel."""
    import helpers as H
    from abpoa_amd import api, synth
    shim = H.cpu_shim_lib()
    sets = [list(synth.make_read_set(137, 0, 4, 34000, 0.03))]
    p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=False, n_threads=4, lib=shim)
    assert dev[0].status == 0 and dev[0].cons_seq == ref[0].cons_seq and dev[0].cons_cov == ref[0].cons_cov


@pytest.mark.parametrize("host", [0, 1], ids=["device_driver", "host_driver"])
def test_extension_mode_goldens_on_ragged_reads(engine, monkeypatch, host):
    """The reference CLI's own output (-m 2, consensus and MSA) for a ragged read-set in which one read hardly aligns at all: its best cell is one base on a
    successor of the source 800 rows down the row order.  The general kernel used to lose the band state of such far successors of the source (it assumed
    a row that enters its look-ahead window untouched unless a LATER row had pushed to it) -- in both drivers; found by tools/fuzz_device_vs_oracle.py."""
    import os
    import helpers as H
    from abpoa_amd import api, seqio
    monkeypatch.setenv("ABPOA_HIP_HOSTGRAPH", str(host))
    D = H.GOLDEN_DIR
    names, seqs, _ = seqio.read_fastx(os.path.join(D, "out_ragged_ext_cons", "input.fa"))
    for name, out_cons, out_msa in (("out_ragged_ext_cons", True, False), ("out_ragged_ext_msa", False, True)):
        r = api.msa_batch([seqs], api.Params(aln_mode=2), out_cons=out_cons, out_msa=out_msa)[0]
        assert api.msa_timing()["n_host_sets"] == host
        assert api.format_output(r, names, out_cons, out_msa) == open(os.path.join(D, name, "output.txt")).read(), name


def test_seeded_fuzz_sweep_of_the_device_driver(engine):
    """200 iterations of tools/fuzz_device_vs_oracle.py (read-set shapes x gap model x alignment mode x band x -s x weights x output kind; reads up to 600
    bases so that the oracle-backed leg stays short) inside the suite: every output of the device-resident driver equals the oracle-backed run's, and the
    sweep reports how many sets left the device and why (abpoa_hip_get_host_reasons) -- a set may leave only for a capacity of the device layout."""
    import importlib.util
    import os
    import time
    import helpers as H
    spec = importlib.util.spec_from_file_location("fuzz_device_vs_oracle", os.path.join(H.ROOT, "tools", "fuzz_device_vs_oracle.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    shim = H.cpu_shim_lib()
    t0 = time.time(); n_dev = n_host = 0; hist = {}
    for it in range(200):
        d, h, _, why = fz.iteration(5 * 100000 + it, shim, small=True)
        n_dev += d; n_host += h
        for k, v in why.items():
            hist[k] = hist.get(k, 0) + v
        if time.time() - t0 > 150:      # (a slow box: what ran is the test)
            break
    print(f"fuzz sweep: {it + 1} iterations in {time.time() - t0:.0f} s, {n_dev} sets on the device, {n_host} through the host driver: {hist}")
    assert set(hist) <= {"edge / aligned slots of a node", "node slots while fusing", "node slots at the first read", "projected graph growth", "cigar slots",
                         "predecessor-list slots", "DP arena too small for the bands"}, hist
    assert n_host <= 0.1 * (n_dev + n_host), (n_dev, n_host, hist)


def test_rows_wider_than_the_band_estimate_stay_on_the_device(engine):
    """A set of three reads in extension mode with linear gaps whose rows are much wider than 2 w (the band pushed off its anchor): the DP arena of the first
    pass is too small, and for a set of a few reads every pass of the ladder has the same node slots (the sum of its reads) -- the later passes grow the
    columns of the arena estimate as well.  Found by tools/fuzz_device_vs_oracle.py --seed 7705 (iteration 103): the set used to end on the host driver."""
    import importlib.util
    import os
    import helpers as H
    spec = importlib.util.spec_from_file_location("fuzz_device_vs_oracle", os.path.join(H.ROOT, "tools", "fuzz_device_vs_oracle.py"))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    d, h, _, why = fz.iteration(770500103, H.cpu_shim_lib())
    assert h == 0 and d == 4, (d, h, why)


def test_ragged_sets_of_a_mixed_job_run_as_a_batch_of_their_own(engine, monkeypatch):
    """A banded global job of uniform read-sets with two ragged ones among them: abpoa_hip_msa_batch hands the ragged sets to the device passes as a batch of
    their own (msa_hip.cpp split_ragged), so the uniform ones keep the all-rounds kernel; results in caller order, equal to the oracle-backed run's and to the
    one-batch form (ABPOA_HIP_NO_RAGGED_SPLIT=1: lock-step launches for everything, no all-rounds launch)."""
    import numpy as np
    import helpers as H
    from abpoa_amd import api, ffi, synth
    shim = H.cpu_shim_lib()
    rng = np.random.default_rng(3)
    sets = []
    for i in range(12):
        reads = list(synth.make_read_set(61, i, 8 + i % 5, 500 + 20 * i, 0.06))
        if i in (2, 9):
            reads = [reads[0]] + [r[int(rng.integers(20, 90)):len(r) - int(rng.integers(20, 90))] for r in reads[1:]]
        sets.append(reads)
    p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
    lib = ffi.lib()
    lib.abpoa_hip_reset_stats()
    apart = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0 and ffi.stats()["rounds_launches"] >= 1
    monkeypatch.setenv("ABPOA_HIP_NO_RAGGED_SPLIT", "1")
    lib.abpoa_hip_reset_stats()
    one = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0 and ffi.stats()["rounds_launches"] == 0
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
    _same(apart, ref, "ragged sets apart")
    _same(one, ref, "one batch")
