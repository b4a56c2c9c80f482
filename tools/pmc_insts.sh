# Instruction counts of the dominant kernel of a bench workload (rocprofv3 PMC, one pass): SQ_INSTS_VALU / SALU / LDS and wave cycles, as a JSON record
# under gpurun_out/ that bench.py reads (copied to profiles/) for its integer-VALU roofline while the row-loop sources are unchanged.
# usage (GPU box): bash tools/pmc_insts.sh cfg2|cfg3|cfg4|cfg5 [read-sets]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg2}; N=${2:-0}
ARGS="--workload $WL --no-cpu-baseline --no-secondary --no-pool --steps 1 --warmup 0"
if [ "$N" != "0" ]; then ARGS="$ARGS --sets $N"; fi
# (one step, no warm-up: a 15 %-error 10 kb job starts at 6x node slots as a warmed-up process does -- the doomed 3x pass is not in the counts)
if [ "$WL" = "cfg3" ]; then export ABPOA_HIP_FIRST_PASS=1; fi
rm -rf /tmp/pmc_out_i /tmp/pmc_out_w
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR -d /tmp/pmc_out_i -o p --output-format csv -- python3 $R/bench.py $ARGS > /tmp/pmc_log_i.txt 2> /tmp/pmc_err_i.txt
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY -d /tmp/pmc_out_w -o p --output-format csv -- python3 $R/bench.py $ARGS > /tmp/pmc_log_w.txt 2> /tmp/pmc_err_w.txt
python3 - "$WL" "$R" <<'PY'
import csv, glob, json, subprocess, sys, collections, os
wl, root = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("/tmp/pmc_out_i", "/tmp/pmc_out_w"):
    f = glob.glob(d + "/**/*counter_collection.csv", recursive=True)[0]
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("abpoa_hip::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
line = json.loads(open("/tmp/pmc_log_i.txt").read().strip().split("\n")[-1])
all_rounds = "poa_rounds_kernel" in line["roofline"]["kernel"]
keys = [k for k in acc if k.startswith(("poa_rounds_kernel",) if all_rounds else ("dp_fast_kernel", "dp_wide_kernel", "dp_local_kernel", "dp_local_team_kernel"))]
tot = collections.defaultdict(float)
for k in keys:
    for c, v in acc[k].items(): tot[c] += v
sys.path.insert(0, root)
import bench
rec = {"workload": wl, "read_sets": line["config"]["read_sets_per_gpu"], "kernel": ", ".join(sorted(keys)), "cells_per_step": line["cells_per_step"],
       "valu_wave_insts_per_step": tot["SQ_INSTS_VALU"], "salu_insts_per_step": tot["SQ_INSTS_SALU"], "lds_insts_per_step": tot["SQ_INSTS_LDS"], "vmem_wr_insts_per_step": tot["SQ_INSTS_VMEM_WR"],
       "wave_cycles_per_step": tot["SQ_WAVE_CYCLES"], "wait_any_per_step": tot["SQ_WAIT_ANY"], "active_inst_any_per_step": tot["SQ_ACTIVE_INST_ANY"],
       "all_kernels": {k: {c: v for c, v in d.items()} for k, d in acc.items() if k.startswith(("dp_", "poa_"))},
       "row_loop_sha": bench.row_loop_sha(),
       "commit": (subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or (open(os.path.join(root, ".git_head")).read().strip() if os.path.exists(os.path.join(root, ".git_head")) else "") or os.environ.get("ABPOA_COMMIT") or "working tree"),
       "how": "tools/pmc_insts.sh: rocprofv3 --kernel-trace --pmc <4 counters>, two passes over `bench.py --workload %s --steps 1 --warmup 0`" % wl}
os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
json.dump(rec, open(os.path.join(root, "gpurun_out", f"{bench.ROUND}_pmc_insts_{wl}.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in rec.items() if k != "all_kernels"}, indent=1))
PY
