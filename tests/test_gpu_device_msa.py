"""GPU (-m gpu): the device-resident read-set driver (graph, fusion, row order, remaining length on the GPU;
abpoa_amd/csrc/poa_device.hip) must give exactly what the host driver gives (which keeps the reference's graph code
and row order, and is itself pinned to the reference's outputs in test_gpu_msa.py / test_host_msa.py):
consensus, per-base coverage and DP cell counts of every set; and, with ABPOA_HIP_DEVSYNC=1, the device graph must
equal the host graph node for node after every read (the library checks that itself and reports on stderr)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _both(sets, params, n_threads=8):
    from abpoa_amd import api
    os.environ["ABPOA_HIP_HOSTGRAPH"] = "1"
    try:
        host = api.msa_batch(sets, params, n_threads=n_threads)
    finally:
        os.environ["ABPOA_HIP_HOSTGRAPH"] = "0"
    dev = api.msa_batch(sets, params, n_threads=n_threads)
    tm = api.msa_timing()
    return host, dev, tm


@pytest.fixture(scope="module")
def engine():
    from abpoa_amd import ffi
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1
    ffi.check(lib.abpoa_hip_init(0))
    return lib


@pytest.mark.parametrize("name,kw,shape", [
    ("affine_5pct", dict(gap_open1=4, gap_open2=0, gap_ext1=2), (48, 16, 400, 0.05)),
    ("affine_15pct", dict(gap_open1=4, gap_open2=0, gap_ext1=2), (24, 20, 300, 0.15)),
    ("convex_default", dict(), (24, 12, 500, 0.10)),
    ("ragged_tiny", dict(gap_open1=4, gap_open2=0, gap_ext1=2), (16, 5, 40, 0.05)),
    # graphs of 1000+ rows: in the all-rounds kernel two wavefronts share the backtrack (backtrack_dir.h)
    ("affine_1kb", dict(gap_open1=4, gap_open2=0, gap_ext1=2), (24, 14, 1000, 0.06)),
    ("convex_1kb_noisy", dict(), (16, 12, 1100, 0.14)),
])
@pytest.mark.parametrize("lockstep", [0, 1], ids=["all_rounds_kernel", "lockstep_rounds"])
def test_device_driver_equals_host_driver(engine, monkeypatch, name, kw, shape, lockstep):
    """Both forms of the device driver: every set through all its rounds in one kernel (poa_rounds.hip; ragged sets: the sets of a batch
    have different numbers of reads), and one launch per phase and round (ABPOA_HIP_LOCKSTEP=1)."""
    from abpoa_amd import api, ffi, synth
    monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", str(lockstep))
    n_sets, n_reads, ln, err = shape
    sets = [synth.make_read_set(11, i, n_reads if i % 5 else max(2, n_reads // 2), ln, err) for i in range(n_sets)]
    engine.abpoa_hip_reset_stats()
    host, dev, tm = _both(sets, api.Params(**kw))
    assert tm["n_groups"] == 1 and tm["n_host_sets"] == 0, f"device driver not used for every set: {tm}"
    assert (ffi.stats()["rounds_launches"] > 0) == (lockstep == 0), ffi.stats()
    for i, (a, b) in enumerate(zip(dev, host)):
        assert a.status == 0 and b.status == 0
        assert a.cons_seq == b.cons_seq, f"{name}: consensus of set {i} differs"
        assert a.cons_cov == b.cons_cov, f"{name}: coverage of set {i} differs"
        assert a.n_cells == b.n_cells, f"{name}: DP cell count of set {i} differs"


@pytest.mark.parametrize("env", [{}, {"ABPOA_HIP_PASS_SETS": "3"}, {"ABPOA_HIP_DIR_WIDE": "1"}, {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"},
                                 {"ABPOA_HIP_DIR_WIDE": "0", "ABPOA_HIP_RING_ROWS": "4"}],
                         ids=["default", "three_passes", "direction_words", "direction_words_ring4", "records_ring4"])
@pytest.mark.parametrize("kw", [dict(), dict(gap_open1=4, gap_open2=0, gap_ext1=2)], ids=["convex", "affine"])
def test_wide_band_jobs_passes_and_arena_formats(engine, monkeypatch, env, kw):
    """Wide-band read-sets (4.5 kb reads: band half-width 55, rows of 3 chunks) through the device-resident driver in the forms the driver picks by
    job size and free memory on big jobs: one pass or several (passes of the resident number of sets), score-record or direction-word arenas, the
    8-row or the 4-row score ring (packed E differences in the convex int32 ring, 4-bit query codes).  All equal to the host driver."""
    from abpoa_amd import api, synth
    sets = [synth.make_read_set(23, i, 6, 4500, 0.08) for i in range(8)]
    host, dev, tm = _both(sets, api.Params(**kw))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev2 = api.msa_batch(sets, api.Params(**kw), n_threads=8)
    assert api.msa_timing()["n_host_sets"] == 0
    for i, (a, b, c) in enumerate(zip(dev, host, dev2)):
        assert a.status == 0 and b.status == 0 and c.status == 0
        assert a.cons_seq == b.cons_seq == c.cons_seq, f"{env}: consensus of set {i} differs"
        assert a.cons_cov == b.cons_cov == c.cons_cov, f"{env}: coverage of set {i} differs"
        assert a.n_cells == b.n_cells == c.n_cells, f"{env}: DP cell count of set {i} differs"


@pytest.mark.parametrize("env", [{}, {"ABPOA_HIP_DIR_WIDE": "1"}, {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"}, {"ABPOA_HIP_NODIR": "1"}],
                         ids=["default", "direction_words", "direction_words_ring4", "records_only"])
@pytest.mark.parametrize("kw", [dict(), dict(gap_open1=4, gap_open2=0, gap_ext1=2)], ids=["convex", "affine"])
def test_mixed_band_widths_in_one_job(engine, monkeypatch, env, kw):
    """One job whose read-sets take different row loops -- 1.5 kb reads (narrow band, one chunk), 3.9 kb (band half-width 49: the wide loop's
    smallest), 7 kb (4 chunks), 11 kb with 12 % error (5-6 chunks, odd read lengths: the 4-bit query packing's last nibble) -- so that narrow and
    wide kernels, both score widths and both arena formats meet in the same launches.  Equal to the host driver."""
    from abpoa_amd import api, synth
    shapes = [(5, 1500, 0.05), (4, 3901, 0.08), (4, 7003, 0.05), (3, 11001, 0.12), (6, 1203, 0.10), (3, 9999, 0.15)]
    sets = [synth.make_read_set(29, i, *shapes[i % len(shapes)]) for i in range(9)]
    host, dev, tm = _both(sets, api.Params(**kw))
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    dev2 = api.msa_batch(sets, api.Params(**kw), n_threads=8)
    assert api.msa_timing()["n_host_sets"] == 0
    for i, (a, b, c) in enumerate(zip(dev, host, dev2)):
        assert a.status == 0 and b.status == 0 and c.status == 0
        assert a.cons_seq == b.cons_seq == c.cons_seq, f"{env}: consensus of set {i} differs"
        assert a.cons_cov == b.cons_cov == c.cons_cov, f"{env}: coverage of set {i} differs"
        assert a.n_cells == b.n_cells == c.n_cells, f"{env}: DP cell count of set {i} differs"


@pytest.mark.parametrize("kw", [dict(), dict(gap_open1=4, gap_open2=0, gap_ext1=2)], ids=["convex", "affine"])
def test_two_wavefronts_on_a_backtrack_change_nothing(engine, monkeypatch, kw):
    """All-rounds kernel, graphs of 800-3000 rows: the backtrack shared by two wavefronts (the helper starts mid-graph and the main walk takes over its
    cigar where the two meet, backtrack_dir.h) against one wavefront per backtrack (ABPOA_HIP_DBG bit 10) and against the host driver: consensus,
    coverage and DP cell counts of 64 read-sets with 3-20 % errors, deletion- and insertion-heavy mixes, 8-25 reads."""
    from abpoa_amd import api, ffi, synth
    shapes = [(8 + (7 * i) % 18, 800 + 137 * (i % 13), 0.03 + 0.017 * (i % 11), None if i % 3 == 0 else ((0.02, 0.09, 0.02) if i % 3 == 1 else (0.02, 0.02, 0.09))) for i in range(64)]
    sets = [synth.make_read_set(41, i, n, ln, err, rates=rt) for i, (n, ln, err, rt) in enumerate(shapes)]
    p = api.Params(**kw)
    engine.abpoa_hip_reset_stats()
    host, two, tm = _both(sets, p)
    assert tm["n_host_sets"] == 0 and ffi.stats()["rounds_launches"] > 0
    monkeypatch.setenv("ABPOA_HIP_DBG", "1024")
    one = api.msa_batch(sets, p, n_threads=8)
    for i, (a, b, c) in enumerate(zip(two, one, host)):
        assert a.status == 0 and b.status == 0 and c.status == 0
        assert a.cons_seq == b.cons_seq == c.cons_seq and a.cons_cov == b.cons_cov == c.cons_cov and a.n_cells == b.n_cells == c.n_cells, f"set {i} {shapes[i]}"


@pytest.mark.parametrize("bt_bytes", ["4096", "6144"])
def test_small_backtrack_windows_with_wide_rows(engine, monkeypatch, bt_bytes):
    """The backtrack's LDS window at its smallest (ABPOA_HIP_BT_BYTES; in the all-rounds kernel four wavefronts share it: about 1 KB each) against noisy
    reads whose bands open to 100+ columns and whose rows keep score records beside their direction words: the window that was copied ahead of the geometry
    loads can then hold no complete row, and the walk has to drop that copy and size the window anew (backtrack_dir.h load_window).  Both forms of the
    driver, equal to the host driver."""
    from abpoa_amd import api, synth
    monkeypatch.setenv("ABPOA_HIP_BT_BYTES", bt_bytes)
    shapes = [(10 + i % 6, 700 + 90 * i, 0.20 + 0.02 * (i % 4), (0.05, 0.10, 0.05) if i % 2 else (0.05, 0.05, 0.10)) for i in range(12)]
    sets = [synth.make_read_set(47, i, n, ln, err, rates=rt) for i, (n, ln, err, rt) in enumerate(shapes)]
    for kw in (dict(gap_open1=4, gap_open2=0, gap_ext1=2), dict()):
        for lockstep in ("0", "1"):
            monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", lockstep)
            host, dev, tm = _both(sets, api.Params(**kw))
            for i, (a, b) in enumerate(zip(dev, host)):
                assert a.status == 0 and b.status == 0, (kw, lockstep, i, a.status, b.status)
                assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov and a.n_cells == b.n_cells, f"{kw} lockstep={lockstep}: set {i} differs"


def test_device_graph_equals_host_graph_after_every_read(engine):
    """Runs in a child process because the check mode is chosen by an environment variable at call time and prints to stderr."""
    code = ("import os,sys; sys.path.insert(0, %r)\n"
            "from abpoa_amd import api, ffi, synth\n"
            "lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))\n"
            "sets = [synth.make_read_set(3, i, 10, 250, 0.08) for i in range(6)]\n"
            "r = api.msa_batch(sets, api.Params(gap_open1=4, gap_open2=0, gap_ext1=2), n_threads=4)\n"
            "print('OK', all(x.status == 0 for x in r), api.msa_timing()['n_host_sets'])\n" % ROOT)
    env = dict(os.environ, ABPOA_HIP_DEVSYNC="1", ABPOA_HIP_HOSTGRAPH="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "OK True 0" in p.stdout
    assert "graph check ok" in p.stderr and "consensus check ok" in p.stderr and "FAILED" not in p.stderr, p.stderr[-3000:]


def test_device_audit_mode_reports_where_every_pool_landed():
    """ABPOA_HIP_DEVICE_AUDIT=1 (child process: the switch is read once): every pool and staging allocation of every device queue is looked up with
    hipPointerGetAttributes and must sit on the device its queue serves, with the queue thread's current device set to it -- a call fails with ENODEV
    otherwise.  Two queues on device 0 here (the pool has one GPU per box); on a multi-GPU node the same switch checks the distinct-ordinal path."""
    code = ("import os,sys; sys.path.insert(0, %r)\n"
            "from abpoa_amd import api, ffi, synth\n"
            "lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))\n"
            "sets = [synth.make_read_set(13, i, 5 + i %% 4, 150 + 30 * (i %% 5), 0.06) for i in range(600)]\n"
            "r = api.msa_batch(sets, api.Params(gap_open1=4, gap_open2=0, gap_ext1=2), n_threads=8)\n"
            "print('OK', all(x.status == 0 for x in r), api.msa_timing()['n_groups'], api.msa_timing()['n_host_sets'])\n" % ROOT)
    env = dict(os.environ, ABPOA_HIP_DEVICE_AUDIT="1", ABPOA_HIP_VERBOSE="1", ABPOA_GPU_DEVICES="0,0", ABPOA_GPU_BATCHES_PER_DEVICE="1")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "OK True 2 0" in p.stdout, (p.stdout, p.stderr[-2000:])
    lines = [ln for ln in p.stderr.splitlines() if "audit:" in ln]
    assert len(lines) >= 8, p.stderr[-2000:]      # (two queues x at least four pools)
    assert all("device 0 (queue device 0, thread's current device 0)" in ln for ln in lines), lines


def test_multi_queue_batch_call_matches_single_queue(engine):
    """abpoa_hip_msa_batch with ABPOA_GPU_DEVICES: cost-sorted batches dealt to per-device queues on worker threads, results in caller
    order.  One GPU here, so the list names device 0 twice (two queues, two pool caches, two streams): same records as the single queue.
    The read-sets are ragged (different read counts and lengths) so that the cost sort really permutes them.
    NOT covered here or anywhere: two DIFFERENT device ordinals (per-device __constant__ argument records of the all-rounds kernel, pool caches on
    several devices) -- the test boxes of this pool have one GPU; the multi-GPU bench of the build driver is the first run of that path."""
    from abpoa_amd import api, synth
    sets = [synth.make_read_set(11, i, 4 + i % 7, 120 + 40 * (i % 5), 0.06) for i in range(600)]
    p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
    one = api.msa_batch(sets, p, n_threads=8)
    os.environ["ABPOA_GPU_DEVICES"] = "0,0"
    os.environ["ABPOA_GPU_BATCHES_PER_DEVICE"] = "2"
    try:
        two = api.msa_batch(sets, p, n_threads=8)
        assert api.msa_timing()["n_groups"] == 2
    finally:
        del os.environ["ABPOA_GPU_DEVICES"], os.environ["ABPOA_GPU_BATCHES_PER_DEVICE"]
    for a, b in zip(one, two):
        assert (a.status, a.cons_seq, a.cons_cov, a.n_cells) == (0, b.cons_seq, b.cons_cov, b.n_cells)


def test_job_shape_hint_skips_the_doomed_pass(engine):
    """Noisy reads outgrow the 3x node estimate of the first device pass; the second such job of the process starts at the factor that held them (4.5x or 6x).  Same results either
    way, equal to the host driver's.  (Child process: the hint is process-wide state and the passes are reported on stderr.)"""
    code = ("import os,sys; sys.path.insert(0, %r)\n"
            "from abpoa_amd import api, ffi, synth\n"
            "lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0))\n"
            "sets = [synth.make_read_set(5, i, 30, 2000, 0.30) for i in range(6)]\n"
            "p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)\n"
            "a = api.msa_batch(sets, p, n_threads=4); sys.stderr.write('SECOND CALL\\n'); b = api.msa_batch(sets, p, n_threads=4)\n"
            "os.environ['ABPOA_HIP_HOSTGRAPH'] = '1'; h = api.msa_batch(sets, p, n_threads=4)\n"
            "print('OK', all(x.status == 0 and x.cons_seq == y.cons_seq == z.cons_seq and x.cons_cov == y.cons_cov == z.cons_cov for x, y, z in zip(a, b, h)))\n" % ROOT)
    env = dict(os.environ, ABPOA_HIP_VERBOSE="1", ABPOA_HIP_HOSTGRAPH="0")
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    assert "OK True" in p.stdout, p.stdout
    first, second = p.stderr.split("SECOND CALL")
    assert "pass 1, node slots 3x" in first and ("pass 2, node slots 4.5x" in first), first[-1500:]
    assert "node slots 3x" not in second and ("node slots 4.5x" in second or "node slots 6x" in second), second[-1500:]


@pytest.mark.parametrize("lockstep", [0, 1], ids=["all_rounds_kernel", "lockstep_rounds"])
def test_device_driver_equals_oracle_backed_host_run(engine, monkeypatch, lockstep):
    """An independent check of the device-resident driver and its row-loop bodies (straight-line copies for 1-8 predecessors and for vectors beyond
    the predecessors' bands, the two-chunk body, the exact bodies): the same read-sets through the CPU build of the host layer whose aligner is the
    plain-C oracle (tests/cpu_shim.cpp).  Shapes chosen to leave the common path: narrow and wide extra bands (-b / -f), 2-25 % errors with
    deletion- and insertion-heavy mixes, convex and affine gaps, ragged read counts, reads of 150-1400 bases."""
    import helpers as H
    from abpoa_amd import api, ffi, synth
    monkeypatch.setenv("ABPOA_HIP_LOCKSTEP", str(lockstep))
    shim = H.cpu_shim_lib()
    cases = [
        (dict(gap_open1=4, gap_open2=0, gap_ext1=2), [(12 + i % 9, 150 + 97 * i, 0.02 + 0.02 * (i % 6), None) for i in range(10)]),
        (dict(), [(10 + i % 7, 300 + 130 * i, 0.12, (0.02, 0.08, 0.02) if i % 2 else (0.02, 0.02, 0.08)) for i in range(8)]),
        (dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=5, extra_f=0.02), [(16, 400 + 150 * i, 0.25, None) for i in range(6)]),
        (dict(gap_open1=6, gap_open2=30, gap_ext1=3, gap_ext2=1, extra_b=30, extra_f=0.03, match=3, mismatch=5), [(9 + i, 500 + 60 * i, 0.10, None) for i in range(6)]),
        # penalties outside the direction words' range (o1 > 7): score-record arenas, on wide bands
        (dict(gap_open1=20, gap_open2=60, gap_ext1=3, gap_ext2=1, extra_b=35, extra_f=0.02), [(7 + i, 600 + 90 * i, 0.12, None) for i in range(5)]),
        (dict(gap_open1=9, gap_open2=0, gap_ext1=4, extra_b=40, extra_f=0.01), [(8, 700 + 50 * i, 0.08, (0.02, 0.03, 0.03)) for i in range(5)]),
    ]
    for kw, shapes in cases:
        sets = [synth.make_read_set(23, i, n, ln, err, rates=rt) for i, (n, ln, err, rt) in enumerate(shapes)]
        p = api.Params(**kw)
        engine.abpoa_hip_reset_stats()
        dev = api.msa_batch(sets, p, n_threads=4)
        assert api.msa_timing()["n_host_sets"] == 0, "device driver not used for every set"
        w_max = p.wb + int(p.wf * max(ln for _, ln, _, _ in shapes))      # (a band half-width of 40 or more takes the wide row loop: one launch per phase and round)
        assert (ffi.stats()["rounds_launches"] > 0) == (lockstep == 0 and w_max < 40)
        ref = api.msa_batch(sets, p, n_threads=4, lib=shim)
        for i, (a, b) in enumerate(zip(dev, ref)):
            assert a.status == 0 and b.status == 0
            assert a.cons_seq == b.cons_seq and a.cons_cov == b.cons_cov, f"{kw}: set {i} differs from the oracle-backed run"


CIGAR_JOBS = {
    # narrow bands, graphs of 700-2500 rows: in the all-rounds kernel four wavefronts share every backtrack (helper parts, merged cigars)
    "narrow_affine": (dict(gap_open1=4, gap_open2=0, gap_ext1=2), [(12, 700 + 60 * i, 0.08) for i in range(8)]),
    "narrow_convex_noisy": (dict(), [(10, 900 + 90 * i, 0.15) for i in range(6)]),
    "narrow_1500": (dict(gap_open1=4, gap_open2=0, gap_ext1=2), [(9, 1500 + 100 * i, 0.05 + 0.02 * i) for i in range(5)]),
    # wide bands (10 kb reads: the all-chunks row loop, column-slice backtrack windows), 5 % and 15 % errors
    "wide_affine_5": (dict(gap_open1=4, gap_open2=0, gap_ext1=2), [(8, 10000 + 17 * i, 0.05) for i in range(4)]),
    "wide_convex_15": (dict(), [(8, 10000 - 23 * i, 0.15) for i in range(4)]),
    # the general kernel's own jobs on the device-resident driver (tests/test_gpu_device_general.py): linear gaps, extension mode, no band
    "linear_banded": (dict(gap_open1=0, gap_open2=0, gap_ext1=2), [(9, 400 + 90 * i, 0.06) for i in range(6)]),
    "extend_convex": (dict(aln_mode=2), [(9, 500 + 70 * i, 0.10) for i in range(6)]),
    "affine_unbanded": (dict(gap_open1=4, gap_open2=0, gap_ext1=2, extra_b=-1), [(8, 300 + 60 * i, 0.08) for i in range(6)]),
}


@pytest.mark.parametrize("job,env", [("narrow_affine", {}), ("narrow_affine", {"ABPOA_HIP_LOCKSTEP": "1"}), ("narrow_convex_noisy", {}), ("narrow_convex_noisy", {"ABPOA_HIP_LOCKSTEP": "1"}),
                                     ("narrow_1500", {}), ("narrow_1500", {"ABPOA_HIP_DBG": "1024"}), ("narrow_1500", {"ABPOA_HIP_DEVICE_GENERAL": "1"}),
                                     ("linear_banded", {}), ("linear_banded", {"ABPOA_HIP_LOCKSTEP": "1"}), ("linear_banded", {"ABPOA_HIP_DEVICE_GENERAL": "1"}), ("extend_convex", {}), ("affine_unbanded", {}),
                                     ("wide_affine_5", {"ABPOA_HIP_DIR_WIDE": "0"}), ("wide_affine_5", {"ABPOA_HIP_DIR_WIDE": "1"}), ("wide_affine_5", {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"}),
                                     ("wide_convex_15", {"ABPOA_HIP_DIR_WIDE": "0"}), ("wide_convex_15", {"ABPOA_HIP_DIR_WIDE": "1"}), ("wide_convex_15", {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"})],
                         ids=lambda v: v if isinstance(v, str) else ("default" if not v else "_".join(f"{k[10:].lower()}{x}" for k, x in v.items())))
def test_device_driver_cigars_equal_the_oracle_backed_run(engine, job, env):
    """Direct cigar comparison (not only consensus / coverage).  With ABPOA_HIP_CIGAR_DIGEST=1 the fuse phase of the device-resident driver folds the graph
    cigar of every alignment into a 64-bit digest per read-set ON THE DEVICE (poa_bodies.h, poa_device.h poa_cigar_digest_round) -- so both of its forms are
    covered: the all-rounds kernel, whose backtrack is shared by four wavefronts (helper cigar parts, merged at the cells where the walks meet), and the
    lock-step launches; narrow bands and 10 kb wide bands in both arena formats (score records / direction words, 8- and 4-row score ring).  The CPU build of
    the host driver, whose aligner is the plain-C oracle, folds the same function over its cigars (msa_batch.cpp): equal digests = equal cigars, word for word."""
    import subprocess
    import sys
    kw, shapes = CIGAR_JOBS[job]
    code = ("import os, sys, ctypes; sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))\n"
            "import helpers as H\n"
            "from abpoa_amd import api, ffi, synth, seqio\n"
            "import numpy as np\n"
            "def digests(lib, sets, m):\n"
            "    out = []\n"
            "    lib.abpoa_hip__cigar_digest.restype = ctypes.c_ulonglong\n"
            "    for s in sets:\n"
            "        a = np.ascontiguousarray(seqio.encode(s[0], m), np.uint8)\n"
            "        out.append(lib.abpoa_hip__cigar_digest(a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), len(a), 0))\n"
            "    return out\n"
            "kw, shapes = %r, %r\n"
            "sets = [synth.make_read_set(23, i, n, ln, err) for i, (n, ln, err) in enumerate(shapes)]\n"
            "p = api.Params(**kw)\n"
            "lib = ffi.lib(); ffi.check(lib.abpoa_hip_init(0)); lib.abpoa_hip_reset_stats()\n"
            "dev = api.msa_batch(sets, p, n_threads=4); tm = api.msa_timing()\n"
            "assert all(r.status == 0 for r in dev) and tm['n_host_sets'] == 0, tm\n"
            "print('ROUNDS_LAUNCHES', ffi.stats()['rounds_launches'])\n"
            "d_dev = digests(lib, sets, p.m); lib.abpoa_hip__cigar_digest(None, 0, 1)\n"
            "shim = H.cpu_shim_lib(); host = api.msa_batch(sets, p, lib=shim, n_threads=8)\n"
            "d_host = digests(shim, sets, p.m); shim.abpoa_hip__cigar_digest(None, 0, 1)\n"
            "assert all(d != 0 for d in d_dev) and d_dev == d_host, (d_dev, d_host)\n"
            "assert [r.cons_seq for r in dev] == [r.cons_seq for r in host]\n"
            "print('CIGARS EQUAL')\n" % (ROOT, ROOT, kw, shapes))
    e = dict(os.environ, ABPOA_HIP_CIGAR_DIGEST="1", **env)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=e, timeout=900)
    assert r.returncode == 0 and "CIGARS EQUAL" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    # (banded linear jobs: the fast row loops and the all-rounds kernel since round 5 -- H records, lane-parallel linear backtrack steps)
    narrow_all_rounds = (job.startswith("narrow") or job == "linear_banded") and "ABPOA_HIP_LOCKSTEP" not in env and "ABPOA_HIP_DEVICE_GENERAL" not in env
    assert ("ROUNDS_LAUNCHES 0" not in r.stdout) == narrow_all_rounds, r.stdout
