# Instruction mix / wave cycles of the row-loop kernels over a whole bench workload (rocprofv3 PMC, one pass per counter group).
# usage (GPU box): bash tools/pmc_bench_rows.sh cfg3 64      (ABPOA_HIP_LOCKSTEP=1 in the environment: one launch per phase and round)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3}; N=${2:-64}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_BRANCH SQ_IFETCH_LEVEL SQ_INSTS_VMEM_RD" "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_out -o p --output-format csv -- python3 $R/bench.py --workload $WL --sets $N --no-cpu-baseline --no-pool --steps 1 --warmup 0 --no-secondary > /tmp/pmc_log.txt 2>/tmp/pmc_err.txt
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" /tmp/pmc_log.txt <<'PY'
import csv, sys, json, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"].split("(")[0].replace("void abpoa_hip::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
line = json.loads(open(sys.argv[2]).read().strip().split("\n")[-1])
print("cells/step", line["cells_per_step"], "sets", line["config"]["read_sets_per_gpu"])
for k, v in acc.items():
    if k.startswith(("dp_", "poa_")) and sum(v.values()) > 1e6: print(k, {c: f"{x:.4g}" for c, x in v.items()})
PY
done
