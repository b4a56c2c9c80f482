"""The C front end of the batch entry (abpoa_amd/host/abpoa_batch.c: the reference's command line, every file of a `-l` list in ONE
abpoa_hip_msa_batch call).  CPU: its own host logic -- option parsing, FASTA / FASTQ / gzip reading, alphabets, matrix files, quality weights,
output text -- linked against the oracle-backed build of the host layer (tests/_build/libcpu_shim.so), compared byte for byte with the
reference's printed outputs under tests/golden/out_* and, where the compiled reference is present (oracle/_ref/abpoa_ref), with a run of it on
the same `-l` list.  GPU (-m gpu): the shipped binary (abpoa_amd/abpoa_batch, linked to libabpoa_hip.so) against the same."""
import gzip
import os
import subprocess

import pytest

import helpers as H
from abpoa_amd import synth

D = H.GOLDEN_DIR
ROOT = H.ROOT
REF = os.path.join(ROOT, "oracle", "_ref", "abpoa_ref")
BLOSUM = os.path.join(D, "data", "BLOSUM62.mtx")


def _golden(name):
    return open(os.path.join(D, name, "output.txt")).read()


def _cpu_binary():
    H.cpu_shim_lib()
    bdir = os.path.join(ROOT, "tests", "_build")
    exe, src = os.path.join(bdir, "abpoa_batch_cpu"), os.path.join(ROOT, "abpoa_amd", "host", "abpoa_batch.c")
    shim = os.path.join(bdir, "libcpu_shim.so")
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(shim)):
        subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-Wall", "-I" + os.path.join(ROOT, "include"), "-o", exe, src, "-L" + bdir, "-lcpu_shim", "-lz", "-lpthread", "-lm",
                               "-Wl,-rpath," + bdir])
    return exe


def _run(exe, args, expect_rc=0):
    p = subprocess.run([exe] + args, capture_output=True, text=True, timeout=600)
    assert p.returncode == expect_rc, (p.returncode, p.stderr[-1500:])
    return p.stdout


CASES = [
    (["-O", "4,0", "-E", "2", os.path.join(D, "data", "seq.fa")], "out_seq_cons"),                    # BASELINE.json configs[0]
    (["-r", "1", os.path.join(D, "data", "test.fa")], "out_test_msa"),
    (["-r", "2", os.path.join(D, "data", "test.fa")], "out_test_cons_msa"),
    ([os.path.join(D, "data", "heter.fa")], "out_heter_cons"),
    (["-O", "4,0", "-E", "2", "-r", "5", os.path.join(D, "data", "seq.fa")], "out_fq_seq"),             # the consensus as FASTQ (reference src/abpoa_output.c:270-276, :516-525)
    (["-r", "5", os.path.join(D, "data", "heter.fa")], "out_fq_heter"),
    (["-m", "1", "-c", "-t", BLOSUM, "-r", "1", os.path.join(D, "aa_blosum_loc", "input.fa")], "aa_blosum_loc"),
    (["-O", "4,0", "-E", "2", "-Q", os.path.join(D, "out_qv_cons", "input.fq")], "out_qv_cons"),
    (["-O", "4,0", "-E", "2", "-Q", "-r", "2", os.path.join(D, "out_qv_msa", "input.fq")], "out_qv_msa"),
    (["-s", "-r", "2", os.path.join(D, "out_rc_msa", "input.fa")], "out_rc_msa"),
    (["-s", "-r", "2", os.path.join(D, "out_rc_long_msa", "input.fa")], "out_rc_long_msa"),
]


def _check_cases(exe):
    for args, name in CASES:
        cmd = open(os.path.join(D, name, "cmd.txt")).read() if os.path.exists(os.path.join(D, name, "cmd.txt")) else ""
        assert _run(exe, args) == _golden(name), f"{name}: {' '.join(args)} (golden made with: {cmd.strip()})"


def _list_job(tmp_path, n=6):
    """a `-l` list of ragged read-sets: FASTA with wrapped lines, one gzip'ed file, nameless records"""
    files = []
    for i in range(n):
        reads = synth.make_read_set(9, i, 4 + i, 150 + 40 * i, 0.06)
        fn = str(tmp_path / f"s{i}.fa")
        with open(fn, "w") as f:
            for j, r in enumerate(reads):
                f.write(f">r{j} some comment\n" if (i + j) % 4 else ">\n")
                for k in range(0, len(r), 60):
                    f.write(r[k:k + 60] + "\n")
        if i == 2:
            with open(fn, "rb") as f, gzip.open(fn + ".gz", "wb") as g:
                g.write(f.read())
            fn += ".gz"
        files.append(fn)
    lst = str(tmp_path / "list.txt")
    open(lst, "w").write("\n".join(files) + "\n")
    return lst


def test_c_front_end_matches_the_reference_outputs():
    _check_cases(_cpu_binary())


def test_c_front_end_list_is_one_batch_and_equals_the_reference_cli(tmp_path):
    exe = _cpu_binary()
    seq = os.path.join(D, "data", "seq.fa"); s1k = os.path.join(D, "out_s1k_cons", "input.fa")
    lst = tmp_path / "two.txt"; lst.write_text(seq + "\n" + s1k + "\n")
    assert _run(exe, ["-O", "4,0", "-E", "2", "-l", str(lst)]) == _golden("out_seq_cons") + _golden("out_s1k_cons")
    if not os.path.exists(REF):
        pytest.skip("compiled reference not present: the golden outputs above are the check")
    job = _list_job(tmp_path)
    # (-z: the z-drop of extension mode, reference src/simd_abpoa_align.c:1018-1026; -e: the end bonus, which the reference parses and nothing in its DP reads)
    for opts in (["-O", "4,0", "-E", "2"], ["-r", "2"], ["-r", "1", "-b", "20", "-f", "0.05"], ["-m", "2", "-z", "10", "-r", "2"], ["-m", "2", "-z", "40", "-e", "7", "-O", "0,0", "-E", "3"],
                 ["-r", "5"], ["-r", "5", "-O", "4,0", "-E", "2", "-m", "1"]):      # (-r 5: the consensus as FASTQ, a quality per base from its coverage, reference src/abpoa_output.c:270-276)
        ref = subprocess.run([REF] + opts + ["-l", job], capture_output=True, text=True, timeout=600)
        assert ref.returncode == 0, (opts, ref.stderr[-300:])
        assert _run(exe, opts + ["-l", job]) == ref.stdout, opts


def _degenerate_job(tmp_path):
    """files the reference treats specially: records without bases in the middle and at the end of a file, a file without records.
    Returns the list, the files, and a twin list whose files lack the empty records (None where a file has none to lose)."""
    files, twins, where = [], [], []
    for i in range(5):
        reads = list(synth.make_read_set(31, i, 5 + i, 90 + 30 * i, 0.08))
        full = list(reads)
        if i == 1:
            full.insert(2, "")
        if i == 2:
            full.append(""); full.insert(1, ""); full.insert(1, "")
        if i == 3:
            full, reads = [], []
        fn, tw = str(tmp_path / f"d{i}.fa"), str(tmp_path / f"t{i}.fa")
        with open(fn, "w") as f:
            f.write("".join(f">q{j}\n{r}\n" if r else f">q{j}\n" for j, r in enumerate(full)))
        with open(tw, "w") as f:
            f.write("".join(f">q{j}\n{r}\n" for j, r in enumerate(full) if r))
        files.append(fn); twins.append(tw); where.append([j for j, r in enumerate(full) if not r])
    lst, lst_t = str(tmp_path / "deg.txt"), str(tmp_path / "twin.txt")
    open(lst, "w").write("\n".join(files) + "\n"); open(lst_t, "w").write("\n".join(twins) + "\n")
    return lst, lst_t, files, twins, where


def _check_degenerate(exe, tmp_path):
    """A record without bases after the first one: the reference aligns the empty query, then abpoa_add_subgraph_alignment reads weight[seq_l - 1] = weight[-1]
    (src/abpoa_graph.c:662) for a source -> sink edge: undefined behaviour, its output depends on what the heap holds.  The engine's rule is the defined part of
    that: the record is an MSA row of gaps and adds nothing to the graph, i.e. the file's output is that of the file without the record, plus the row.
    Deterministic in the reference and compared with it: a file without records prints nothing; a FIRST record without bases ends the run."""
    lst, lst_t, files, twins, where = _degenerate_job(tmp_path)
    for opts in ([], ["-r", "1"], ["-r", "2"], ["-O", "4,0", "-E", "2", "-r", "2"]):
        got = _run(exe, opts + ["-l", lst])
        assert _run(exe, opts + ["-l", lst, "--piece", "2", "--readers", "3"]) == got, ("pieces of two files", opts)
        want = ""
        for fn, tw, gaps in zip(files, twins, where):
            ref = subprocess.run([REF] + opts + [tw], capture_output=True, text=True, timeout=600)
            assert ref.returncode == 0, ref.stderr[-500:]
            lines = ref.stdout.splitlines()
            if gaps and "-r" in opts and lines:      # MSA rows: put the rows of gaps where the empty records were
                width = len(lines[1]); recs = [lines[k:k + 2] for k in range(0, len(lines), 2)]
                cons = [r for r in recs if r[0] == ">Consensus_sequence"]; rows = [r for r in recs if r[0] != ">Consensus_sequence"]
                for g in gaps:
                    rows.insert(g, [f">q{g}", "-" * width])
                lines = [x for r in rows + cons for x in r]
            want += "".join(x + "\n" for x in lines)
        assert got == want, opts
    # a first record without bases: the reference dies in abpoa_add_graph_sequence (single file, or the first of a list).  Later in a list the reference
    # aligns the bases the PREVIOUS file had at that record position (its abpoa_seq_t is not cleared between files: abpoa_cpy_str, src/abpoa_seq.c:123-130) --
    # a state leak, not reproduced: the engine ends the run there too, after the output of the files before it
    bad = str(tmp_path / "first_empty.fa"); open(bad, "w").write(">a\n>b\nACGTACGTAGCTAGCTAGCATCGATCGATGCA\n>c\nACGTACGTAGCTAGCTAGCATCGTCGATGCA\n")
    ref = subprocess.run([REF, bad], capture_output=True, text=True, timeout=600)
    p = subprocess.run([exe, bad], capture_output=True, text=True, timeout=600)
    assert ref.returncode != 0 and p.returncode == 1 and p.stdout == ref.stdout == "" and "seq_l: 0" in p.stderr and "seq_l: 0" in ref.stderr
    lst2 = str(tmp_path / "deg2.txt"); open(lst2, "w").write(files[0] + "\n" + files[3] + "\n" + bad + "\n" + files[4] + "\n")
    p = subprocess.run([exe, "-l", lst2], capture_output=True, text=True, timeout=600)
    assert p.returncode == 1 and p.stdout == _run(exe, [files[0]]) and "seq_l: 0" in p.stderr


def test_c_front_end_degenerate_inputs_equal_the_reference(tmp_path):
    if not os.path.exists(REF):
        pytest.skip("compiled reference not present")
    _check_degenerate(_cpu_binary(), tmp_path)


def test_c_front_end_streams_the_list_in_pieces(tmp_path):
    """`-l` is streamed: pieces of --piece files, the next piece read by --readers threads while the current one is in the batch call; the output is the
    same text in list order whatever the piece size (sticky record names cross piece borders like they cross files)."""
    exe = _cpu_binary()
    job = _list_job(tmp_path, 9)
    whole = _run(exe, ["-r", "2", "-l", job, "--piece", "100"])
    for piece, readers in ((1, 1), (2, 4), (4, 2), (9, 8)):
        assert _run(exe, ["-r", "2", "-l", job, "--piece", str(piece), "--readers", str(readers)]) == whole, (piece, readers)
    if os.path.exists(REF):
        ref = subprocess.run([REF, "-r", "2", "-l", job], capture_output=True, text=True, timeout=600)
        assert ref.returncode == 0 and ref.stdout == whole


def test_c_front_end_refuses_what_the_engine_does_not_build():
    exe = _cpu_binary()
    seq = os.path.join(D, "data", "seq.fa")
    for bad in (["-r", "3"], ["-r", "4"], ["-S"], ["-d", "2"], ["-p"]):
        p = subprocess.run([exe] + bad + [seq], capture_output=True, text=True, timeout=60)
        assert p.returncode == 2 and "outside this engine" in p.stderr and p.stdout == ""


@pytest.mark.gpu
def test_shipped_binary_on_the_gpu(tmp_path):
    """abpoa_amd/abpoa_batch (C, linked to libabpoa_hip.so only): golden outputs, and a `-l` list of 40 read-sets against the compiled reference
    (oracle/_ref/abpoa_ref travels to the GPU box prebuilt), consensus and MSA."""
    exe = os.path.join(ROOT, "abpoa_amd", "abpoa_batch")
    assert os.path.exists(exe), "abpoa_amd/abpoa_batch not built (make -C abpoa_amd/csrc)"
    _check_cases(exe)
    if os.path.exists(REF):
        job = _list_job(tmp_path, 40)
        for opts in (["-O", "4,0", "-E", "2"], ["-r", "2"], ["-O", "4,0", "-E", "2", "-r", "1"]):
            ref = subprocess.run([REF] + opts + ["-l", job], capture_output=True, text=True, timeout=600)
            assert ref.returncode == 0
            assert _run(exe, opts + ["-l", job]) == ref.stdout, opts
            assert _run(exe, opts + ["-l", job, "--piece", "7"]) == ref.stdout, ("streamed in pieces of 7 files", opts)
        _check_degenerate(exe, tmp_path)


@pytest.mark.gpu
def test_shipped_binary_against_the_reference_on_ragged_reads_and_every_mode(tmp_path):
    """The compiled reference itself (oracle/_ref/abpoa_ref, prebuilt; not the oracle restatement) against the C front end on the jobs the device-resident
    driver took over in round 4: read-sets whose reads are cut at random places (the source / sink collect dozens of edges, the band anchor sits off the path),
    under global / local / extension alignment, linear / affine / convex gaps, no band, and -s on sets with reverse-complemented reads -- byte for byte."""
    import numpy as np
    exe = os.path.join(ROOT, "abpoa_amd", "abpoa_batch")
    assert os.path.exists(exe), "abpoa_amd/abpoa_batch not built (make -C abpoa_amd/csrc)"
    if not os.path.exists(REF):
        pytest.skip("the compiled reference is not on this box")
    rng = np.random.default_rng(77)
    comp = str.maketrans("ACGT", "TGCA")
    files, files_rc = [], []
    for i in range(16):
        reads = list(synth.make_read_set(19, i, 8 + 3 * i, 120 + 55 * i, 0.03 + 0.01 * (i % 6)))
        cut = [reads[0]]
        for r in reads[1:]:
            a = int(rng.integers(0, int(0.15 * len(r)) + 1)); b = len(r) - int(rng.integers(0, int(0.15 * len(r)) + 1))
            cut.append(r[a:b])
        for kind, rs, lst in (("cut", cut, files), ("rc", [(r[::-1].translate(comp) if (j and j % 3 == 0) else r) for j, r in enumerate(cut)], files_rc)):
            fn = str(tmp_path / f"{kind}{i}.fa")
            with open(fn, "w") as f:
                f.write("".join(f">r{j}\n{r}\n" for j, r in enumerate(rs)))
            lst.append(fn)
    job, job_rc = str(tmp_path / "cut.txt"), str(tmp_path / "rc.txt")
    open(job, "w").write("\n".join(files) + "\n"); open(job_rc, "w").write("\n".join(files_rc) + "\n")
    cases = [(job, []), (job, ["-r", "2"]), (job, ["-O", "4,0", "-E", "2", "-r", "1"]), (job, ["-m", "1", "-r", "2"]), (job, ["-m", "2", "-r", "1"]),
             (job, ["-O", "0,0", "-E", "2", "-r", "2"]), (job, ["-b", "-1", "-r", "1"])]
    for lst, opts in cases:
        ref = subprocess.run([REF] + opts + ["-l", lst], capture_output=True, text=True, timeout=900)
        assert ref.returncode == 0, (opts, ref.stderr[-500:])
        assert _run(exe, opts + ["-l", lst]) == ref.stdout, opts
    # -s: file by file -- the reference's own `-s -l` run over several files ends in "free(): double free" (its strand flags outlive a file: abpoa_poa frees
    # the query of a read whose flag an earlier file left set) -- while the batch binary takes the whole list in one call: its output is the concatenation
    for opts in (["-s", "-r", "2"], ["-s", "-O", "4,0", "-E", "2"]):
        want = ""
        for fn in files_rc:
            ref = subprocess.run([REF] + opts + [fn], capture_output=True, text=True, timeout=900)
            assert ref.returncode == 0, (opts, fn, ref.stderr[-500:])
            want += ref.stdout
            assert _run(exe, opts + [fn]) == ref.stdout, (opts, fn)
        assert job_rc and _run(exe, opts + ["-l", job_rc]) == want, opts


def test_host_layer_under_address_and_ub_sanitizers(tmp_path):
    """the host layer (graph fusion, Kahn order, consensus, RC-MSA in poa_graph.cpp; the threaded batch driver msa_batch.cpp; the C front
    end) built with -fsanitize=address,undefined together with the oracle and run on golden inputs and on a threaded `-l` job (two groups of read-sets
    advancing on their own threads): same outputs, no report.  (CPU build only: GPU sanitizers are not available on this pool.)"""
    bdir = os.path.join(ROOT, "tests", "_build", "asan")
    os.makedirs(bdir, exist_ok=True)
    exe = os.path.join(bdir, "abpoa_batch_asan")
    csrc, odir = os.path.join(ROOT, "abpoa_amd", "csrc"), os.path.join(ROOT, "oracle")
    srcs_cpp = [os.path.join(ROOT, "tests", "cpu_shim.cpp"), os.path.join(csrc, "poa_graph.cpp"), os.path.join(csrc, "msa_batch.cpp"), os.path.join(csrc, "engine_options.cpp")]
    srcs_c = [os.path.join(odir, "abpoa_dp_oracle.c"), os.path.join(odir, "dir_model.c"), os.path.join(ROOT, "abpoa_amd", "host", "abpoa_batch.c")]
    deps = srcs_cpp + srcs_c + [os.path.join(csrc, "poa_graph.h"), os.path.join(csrc, "msa_batch.h"), os.path.join(ROOT, "include", "abpoa_hip.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
        san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-g", "-O1"]
        objs = []
        for f in srcs_c:
            objs.append(os.path.join(bdir, os.path.basename(f) + ".o"))
            subprocess.check_call(["gcc", "-std=gnu99"] + san + ["-c", "-I" + os.path.join(ROOT, "include"), "-o", objs[-1], f])
        for f in srcs_cpp:
            objs.append(os.path.join(bdir, os.path.basename(f) + ".o"))
            subprocess.check_call(["g++", "-std=c++17"] + san + ["-c", "-I" + os.path.join(ROOT, "include"), "-o", objs[-1], f])
        subprocess.check_call(["g++"] + san + ["-pthread", "-o", exe] + objs + ["-lz"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    job = _list_job(tmp_path, 8)
    runs = [(["-O", "4,0", "-E", "2", os.path.join(D, "data", "seq.fa")], _golden("out_seq_cons")),
            (["-r", "2", os.path.join(D, "data", "test.fa")], _golden("out_test_cons_msa")),
            (["-m", "1", "-c", "-t", BLOSUM, "-r", "1", os.path.join(D, "aa_blosum_loc", "input.fa")], _golden("aa_blosum_loc")),
            (["-s", "-r", "2", os.path.join(D, "out_rc_msa", "input.fa")], _golden("out_rc_msa")),
            (["-r", "2", "-T", "4", "-l", job], None)]
    plain = _cpu_binary()
    for args, want in runs:
        p = subprocess.run([exe] + args, capture_output=True, text=True, env=env, timeout=900)
        assert p.returncode == 0 and "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, (args, p.stderr[-3000:])
        assert p.stdout == (want if want is not None else _run(plain, args)), args
