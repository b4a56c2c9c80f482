"""GPU (-m gpu): the HIP engine, called through the C-ABI (abpoa_hip_align_batch), must equal
 (a) the committed golden vectors from the compiled reference and
 (b) the C oracle on the same inputs -- bands, every score-plane cell, row arg-max, best score, cigar,
     abpoa_res_t fields and the max_pos_left/right state, bit for bit."""
import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu
CASES = H.golden_cases()


@pytest.fixture(scope="module")
def engine():
    from abpoa_amd import ffi
    lib = ffi.lib()
    assert lib.abpoa_hip_device_count() >= 1, "no HIP device: the engine has no fallback"
    ffi.check(lib.abpoa_hip_init(0))
    return lib


@pytest.mark.parametrize("label,path", CASES, ids=[c[0] for c in CASES])
def test_hip_matches_golden_and_oracle(engine, label, path):
    g = H.read_abpg(path)
    case = H.FlatCase(g)
    o = H.run_oracle(case)
    h = H.run_hip([case])[0]
    H.compare_with_golden(h, g, label="hip-vs-golden " + label)
    H.compare_outs(h, o, label="hip-vs-oracle " + label)
    # without the trace the fast row loops write direction words instead of score records (dir_plane.h) and the backtrack walks those: best cell, cigar,
    # every abpoa_res_t field and the band state must not change
    _dir_counts(engine)
    h2 = H.run_hip([case], want_trace=False)[0]
    H.compare_with_golden(h2, g, check_planes=False, label="hip-dir-vs-golden " + label)
    assert np.array_equal(h2.cigar, o.cigar) and h2.best_score == o.best_score
    walked, redone = _dir_counts(engine)
    if label.startswith(("s1k_", "seq_ag_gb", "heter_ag_gb", "heter_cg_gb")):      # global, narrow band, default penalties: the plane must have been used
        assert walked == 1 and redone == 0, (label, walked, redone)


@pytest.mark.parametrize("env", [{}, {"ABPOA_HIP_NOWIDE": "1"}, {"ABPOA_HIP_DIR_WIDE": "1"}, {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"}],
                         ids=["default", "nowide", "dir_wide", "dir_wide_ring4"])
def test_direction_words_match_the_model(engine, monkeypatch, env):
    """Plane level: every direction word the row loops write (trace mode with ABPOA_HIP_DIRTRACE=1) carries the decisions oracle/dir_model.c derives from
    the oracle's scores for that cell.  With ABPOA_HIP_NOWIDE=1 the 10 kb goldens run through the narrow kernel's chunk-by-chunk bodies, which write
    the words too (their wide row loop keeps score records); with ABPOA_HIP_DIR_WIDE=1 the all-chunks wide row loop writes them (what the device-resident
    driver switches to when the record arenas of a 10 kb job do not fit the device), also with a 4-row score ring (every other row keeps its records for a reader in HBM)."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 0
    for label, path in CASES:
        if not label.startswith(("s1k_", "seq_ag_gb", "heter_ag_gb", "heter_cg_gb") + (("s10k_",) if env else ())):
            continue
        g = H.read_abpg(path)
        bad = H.dir_words_mismatches(H.FlatCase(g), g)
        assert bad is not None and bad == [], (label, bad[:5] if bad else bad)
        n += 1
    assert n >= 4


@pytest.mark.parametrize("e1,wb", [(2, 10), (1, 3), (4, 2), (3, 6)], ids=["e2_b10", "e1_b3", "e4_b2", "e3_b6"])
def test_linear_rows_plane_level_on_the_golden_graphs(engine, e1, wb):
    """Linear gaps on the fast row loops (rows_fast.h GAP = 0) at plane level: the graphs and queries of the banded global goldens, re-scored with linear gaps
    (reference simd_abpoa_lg_dp, src/simd_abpoa_align.c:701-779) and narrow bands -b 2 .. 10 -f 0 -- bands so tight that stretches of a row are reached by no
    real score, where the reference's `max(H, first)` clamps every lane at `inf` (tests/test_linear_closed_form.py) -- every H cell, band, row arg-max, best
    score, cigar and the max_pos_left/right state against the C oracle."""
    n = 0
    cases = []
    for label, path in CASES:
        if not label.split("/")[0].endswith("_gb"):
            continue
        g = dict(H.read_abpg(path))
        if int(np.asarray(g["align_mode"]).reshape(-1)[0]) != 0 or int(np.asarray(g["wb"]).reshape(-1)[0]) < 0:
            continue
        g["gap_mode"] = np.array([0], np.int32); g["gap_open1"] = np.array([0], np.int32); g["gap_open2"] = np.array([0], np.int32)
        g["gap_ext1"] = np.array([e1], np.int32); g["wb"] = np.array([wb], np.int32); g["wf"] = np.array([0.0], np.float32)
        cases.append((label, H.FlatCase(g)))
    assert len(cases) >= 10
    outs = []
    for i in range(0, len(cases), 16):
        grp = [c for _, c in cases[i:i + 16] if c.sc.m == cases[i][1].sc.m]
        if len(grp) != len(cases[i:i + 16]):      # (run_hip takes one alphabet per call)
            for _, c in cases[i:i + 16]:
                outs.append(H.run_hip([c])[0])
        else:
            outs.extend(H.run_hip(grp))
    for (label, case), h in zip(cases, outs):
        H.compare_outs(h, H.run_oracle(case), label=f"linear e={e1} b={wb} hip-vs-oracle {label}")
        n += 1
    assert n == len(cases)


@pytest.mark.parametrize("team", ["0", "1"], ids=["one_wavefront", "team_of_four"])
def test_local_row_loops_plane_level(engine, monkeypatch, team):
    """Both forms of the local row loop (rows_local.h) -- one wavefront with every chunk of a row in registers, and the team of four wavefronts that split
    the chunks (one LDS exchange of carry-chain results and arg-max keys per row) -- on the local goldens (nucleotides, affine / convex; amino acids with
    BLOSUM62) and on synthetic local alignments of 3-9 chunks with ragged chunk shares (reads of 150-560 residues; graphs with 1-6 predecessors per row,
    predecessors older than the score ring): every plane cell, best cell, cigar against the golden vectors and the oracle."""
    import os
    from abpoa_amd import api, synth, workloads
    monkeypatch.setenv("ABPOA_HIP_LOCAL_TEAM", team)
    n = 0
    for label, path in CASES:
        if "_loc" not in label:
            continue
        g = H.read_abpg(path)
        case = H.FlatCase(g)
        h = H.run_hip([case])[0]
        H.compare_with_golden(h, g, label=f"local team={team} {label}")
        H.compare_outs(h, H.run_oracle(case), label=f"local team={team} vs oracle {label}")
        n += 1
    assert n >= 3
    # whole MSAs (30 rounds of growing graphs) through the device driver: 150 .. 560 residues = 3 .. 9 chunks, 1 / 2 / 3 chunks per wavefront with uneven shares
    shim = H.cpu_shim_lib()
    p = api.Params(aln_mode=1, is_aa=True, score_matrix=workloads.BLOSUM62)
    sets = [synth.make_read_set(71, i, 14, ln, alphabet=synth.AA, rates=(0.07, 0.03, 0.03)) for i, ln in enumerate((150, 200, 260, 330, 390, 450, 515, 560))]
    dev = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4)
    assert api.msa_timing()["n_host_sets"] == 0
    ref = api.msa_batch(sets, p, out_cons=True, out_msa=True, n_threads=4, lib=shim)
    for i, (a, b) in enumerate(zip(dev, ref)):
        assert a.status == 0 and a.msa_seq == b.msa_seq and a.cons_seq == b.cons_seq and a.n_cells == b.n_cells, f"team={team}: set {i}"


@pytest.mark.parametrize("bt_bytes", ["8192", "12288", "28672"])
@pytest.mark.parametrize("nodir", ["0", "1"], ids=["words", "records"])
def test_wide_loop_on_narrow_w_alignments_with_small_windows(engine, monkeypatch, bt_bytes, nodir):
    """Goldens ragged_ag_gb (a short read against a longer graph: rows ~150-250 columns wide although w = 19) through the WIDE row loop (ABPOA_HIP_WIDE_LO=10:
    what the device-resident driver does for read-sets with ragged ends) and the tail's column-slice windows of 8 / 12 / 28 KB.  int16-affine records are 8 bytes
    and their slices two columns wider than planned (even-column rounding): the window used to be sized without that and its topmost rows read back zeros."""
    monkeypatch.setenv("ABPOA_HIP_WIDE_LO", "10")
    monkeypatch.setenv("ABPOA_HIP_BT_BYTES", bt_bytes)
    monkeypatch.setenv("ABPOA_HIP_NODIR", nodir)
    n = 0
    for label, path in CASES:
        if not label.startswith("ragged_ag_gb"):
            continue
        g = H.read_abpg(path)
        case = H.FlatCase(g)
        h = H.run_hip([case], want_trace=False)[0]
        assert h.status == 0 and h.n_cigar == len(g["cigar"]) and (h.cigar == g["cigar"]).all() and h.best_score == int(g["best_score"][0]), f"{label} window {bt_bytes}"
        n += 1
    assert n == 2


def test_need_scores_redo_path(engine, monkeypatch):
    """The direction-plane walk gives up (ABPOA_HIP_STATUS_NEED_SCORES) where the words cannot decide where an F value came from -- never seen on real
    data -- and the alignment is redone with score records.  ABPOA_HIP_DBG=512 makes the walk give up at the first insertion it would decide from
    the words, so that the redo path runs: same results, and the counters say that alignments were redone."""
    monkeypatch.setenv("ABPOA_HIP_DBG", "512")
    engine.abpoa_hip_reset_stats()
    _dir_counts(engine)
    n = redone_total = 0
    for label, path in CASES:
        if not label.startswith(("s1k_", "heter_ag_gb", "heter_cg_gb")):
            continue
        g = H.read_abpg(path)
        h = H.run_hip([H.FlatCase(g)], want_trace=False)[0]
        H.compare_with_golden(h, g, check_planes=False, label=f"need-scores {label}")
        walked, redone = _dir_counts(engine)
        redone_total += redone
        n += 1
    assert n >= 3 and redone_total >= 1, (n, redone_total)
    # the device-resident driver hands such a set to the host driver, which redoes the alignment with records
    from abpoa_amd import api, synth
    sets = [synth.make_read_set(31, i, 8, 600, 0.12) for i in range(6)]
    p = api.Params(gap_open1=4, gap_open2=0, gap_ext1=2)
    dev = api.msa_batch(sets, p, n_threads=4)
    monkeypatch.setenv("ABPOA_HIP_DBG", "0")
    ref = api.msa_batch(sets, p, n_threads=4)
    for a, b_ in zip(dev, ref):
        assert a.status == 0 and b_.status == 0 and a.cons_seq == b_.cons_seq and a.cons_cov == b_.cons_cov


def _dir_counts(lib):
    import ctypes
    out = (ctypes.c_longlong * 2)()
    lib.abpoa_hip__dir_counts(out)
    return out[0], out[1]


def test_hip_batch_mixed_widths(engine):
    """One launch with int16 and int32 alignments side by side (same scoring), order preserved."""
    import os
    groups = {}
    for label, path in CASES:
        g = H.read_abpg(path)
        key = (int(g["m"][0]), int(g["gap_mode"][0]), int(g["align_mode"][0]), int(g["wb"][0]), int(g["zdrop"][0]),
               tuple(g["mat"].tolist()), int(g["gap_open1"][0]), int(g["gap_ext1"][0]), int(g["gap_open2"][0]), int(g["gap_ext2"][0]))
        groups.setdefault(key, []).append((label, g))
    nbatch = 0
    for key, items in groups.items():
        if len(items) < 2:
            continue
        cases = [H.FlatCase(g) for _, g in items]
        outs = H.run_hip(cases, want_trace=False)
        for (label, g), o in zip(items, outs):
            H.compare_with_golden(o, g, check_planes=False, label="batch " + label)
        nbatch += 1
    assert nbatch >= 3


def test_arena_overflow_retry_keeps_traces(engine, monkeypatch):
    """Batches whose arenas are under-sized on purpose (ABPOA_HIP_ARENA_PCT): alignments that overflow are re-run at full width in a second
    pass that re-uses the device arenas; the planes of the alignments that finished in the first pass must still come back intact."""
    monkeypatch.setenv("ABPOA_HIP_ARENA_PCT", "55")
    groups = {}
    for label, path in CASES:
        g = H.read_abpg(path)
        if int(g["wb"][0]) < 0 or int(g["align_mode"][0]) != 0 or "row_checksum" in g and len(g.get("planes", [])) == 0:
            continue
        key = (int(g["m"][0]), int(g["gap_mode"][0]), int(g["zdrop"][0]), tuple(g["mat"].tolist()), int(g["gap_open1"][0]), int(g["gap_ext1"][0]),
               int(g["gap_open2"][0]), int(g["gap_ext2"][0]))
        groups.setdefault(key, []).append((label, g))
    n = 0
    for key, items in groups.items():
        if len(items) < 2:
            continue
        outs = H.run_hip([H.FlatCase(g) for _, g in items], want_trace=True)
        for (label, g), o in zip(items, outs):
            H.compare_with_golden(o, g, label="overflow-retry " + label)
        n += 1
    assert n >= 2


@pytest.mark.parametrize("env", [{"ABPOA_HIP_RING_ROWS": "4"}, {"ABPOA_HIP_TEAM": "2"}, {"ABPOA_HIP_TEAM": "4"}, {"ABPOA_HIP_NOWIDE": "1"}, {"ABPOA_HIP_DIR_WIDE": "1"},
                                 {"ABPOA_HIP_DIR_WIDE": "1", "ABPOA_HIP_RING_ROWS": "4"}],
                         ids=["ring4_hbm_gather", "team2", "team4", "nowide", "dir_wide", "dir_wide_ring4"])
def test_wide_band_variants(engine, monkeypatch, env):
    """The 10 kb goldens (4-5 chunks per row) through the other forms of the wide row loop: a 4-row score ring (every other row gathers a predecessor
    from the HBM arena), teams of 2 / 4 wavefronts per alignment, the chunk-by-chunk loop of the narrow kernel, and the wide loop with direction words."""
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n = 0
    for label, path in CASES:
        if not label.startswith(("s10k_", "s20k_")):
            continue
        g = H.read_abpg(path)
        h = H.run_hip([H.FlatCase(g)])[0]
        H.compare_with_golden(h, g, label=f"{env} {label}")
        h2 = H.run_hip([H.FlatCase(g)], want_trace=False)[0]      # no trace: direction-plane arenas where the launch uses them (narrow-band row loop: the nowide variant)
        H.compare_with_golden(h2, g, check_planes=False, label=f"{env} dir {label}")
        n += 1
    assert n >= 3
    from abpoa_amd import api, synth, workloads as W
    for wl in ("cfg3", "cfg4"):
        w = W.WORKLOADS[wl]
        r = api.msa_batch([synth.make_read_set(1, 2, **synth.CONFIGS[w["cfg"]])], api.Params(**w["params"]))[0]
        assert r.status == 0 and W.output_sha(api.format_output(r)) == W.load_digests(wl)[2], f"{env} {wl} set 2"
