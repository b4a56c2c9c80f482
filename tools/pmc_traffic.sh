# HBM traffic of the row-loop kernels on a bench workload: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in SEPARATE passes (they do not fit one
# pass; no tracing flag besides --kernel-trace), the same bench command each time.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts
# 128-byte requests at 64 bytes -> x2; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Writes profiles/<round>_pmc_traffic_<workload>.json,
# which bench.py reports as roofline.traffic while the row-loop sources are unchanged (hash kept in the record).
# usage (on the GPU box): bash tools/pmc_traffic.sh cfg2|cfg3|cfg4|cfg5 [read-sets]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg2}; N=${2:-0}
ARGS="--workload $WL --no-cpu-baseline --no-secondary --no-pool --steps 1 --warmup 0"
if [ "$N" != "0" ]; then ARGS="$ARGS --sets $N"; fi
# (one step, no warm-up: a 15 %-error 10 kb job starts at 6x node slots as a warmed-up process does -- the doomed 3x pass is not in the counts)
if [ "$WL" = "cfg3" ]; then export ABPOA_HIP_FIRST_PASS=1; fi
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_out_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_out_$c -o p --output-format csv -- python3 $R/bench.py $ARGS > /tmp/pmc_log_$c.txt 2> /tmp/pmc_err_$c.txt
done
python3 - "$WL" "$R" <<'PY'
import csv, glob, json, subprocess, sys, collections, hashlib, os
wl, root = sys.argv[1], sys.argv[2]
tot = {}; per_kernel = collections.defaultdict(lambda: collections.defaultdict(float)); launches = collections.Counter()
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_out_{c}/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c: continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("abpoa_hip::", "")
        per_kernel[k][c] += float(r["Counter_Value"])
        if c == "FETCH_SIZE": launches[k] += 1
line = json.loads(open("/tmp/pmc_log_WRITE_SIZE.txt").read().strip().split("\n")[-1])
rounds = line["roofline"]["launches"]
# the kernel the bench line's roofline is about: the all-rounds kernel where the job took it (narrow bands), else the row-loop kernels of a round
all_rounds = "poa_rounds_kernel" in line["roofline"]["kernel"]
rows = {k: v for k, v in per_kernel.items() if k.startswith(("poa_rounds_kernel",) if all_rounds else ("dp_fast_kernel", "dp_wide_kernel", "dp_team_kernel", "dp_local_kernel", "dp_local_team_kernel"))}
fetch_kb = sum(v["FETCH_SIZE"] for v in rows.values()); write_kb = sum(v["WRITE_SIZE"] for v in rows.values())
hbm = (2 * fetch_kb + write_kb) * 1024 / max(1, rounds)
steps = max(1, line["steps"])
sys.path.insert(0, root)
import bench
rec = {"workload": wl, "read_sets": line["config"]["read_sets_per_gpu"], "rounds": rounds,
       "hbm_bytes_per_launch": int(hbm), "algo_bytes_per_launch": line["roofline"]["algo_bytes_per_launch"],
       # per STEP (one pass over the batch): what a bench run of the same workload can be compared with whatever its launch count
       "hbm_bytes_per_step": int((2 * fetch_kb + write_kb) * 1024 / steps), "fetch_bytes_per_step": int(2 * fetch_kb * 1024 / steps), "write_bytes_per_step": int(write_kb * 1024 / steps),
       "algo_bytes_per_step": int(line["roofline"]["algo_bytes_per_launch"] * rounds / steps),
       "hbm_over_algorithmic": round((2 * fetch_kb + write_kb) * 1024 / max(1.0, line["roofline"]["algo_bytes_per_launch"] * rounds), 3),
       "FETCH_SIZE_KB_raw_per_launch": round(fetch_kb / max(1, rounds), 1), "WRITE_SIZE_KB_per_launch": round(write_kb / max(1, rounds), 1),
       "fetch_correction": "x2 on gfx950 (MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 128-B requests at 64 B)",
       "launch": ("one launch of abpoa_hip::poa_rounds_kernel = rounds 2..n of every read-set (graph phases, row loop and backtrack inside: its traffic includes the backtrack's re-read of the arenas)"
                  if all_rounds else "one round of the progressive alignment = the row-loop kernels of that round (all score widths)"),
       "row_loop_sha": bench.row_loop_sha(),
       "commit": (subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or (open(os.path.join(root, ".git_head")).read().strip() if os.path.exists(os.path.join(root, ".git_head")) else "") or os.environ.get("ABPOA_COMMIT") or "working tree"),
       "all_kernels_MB_per_launch": {k: {"fetch_x2": round(2 * v["FETCH_SIZE"] / 1024 / max(1, rounds), 1), "write": round(v["WRITE_SIZE"] / 1024 / max(1, rounds), 1)} for k, v in per_kernel.items()},
       "how": "tools/pmc_traffic.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --workload %s --steps 1 --warmup 0`" % wl}
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(root, "gpurun_out", f"{bench.ROUND}_pmc_traffic_{wl}.json"), "w"), indent=1)
print(json.dumps(rec, indent=1))
PY
