# Instruction-cache behaviour of the row-loop kernels of a bench workload (rocprofv3 PMC, one pass): SQC_ICACHE_REQ / HITS / MISSES, SQ_IFETCH.
# usage (GPU box): bash tools/pmc_icache.sh cfg3|cfg4 [read-sets]   (environment switches such as ABPOA_HIP_NODIR=1 are inherited)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg3}; N=${2:-256}
if [ "$WL" = "cfg3" ]; then export ABPOA_HIP_FIRST_PASS=1; fi
rm -rf /tmp/pmc_out_ic
rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d /tmp/pmc_out_ic -o p --output-format csv -- python3 $R/bench.py --workload $WL --sets $N --no-cpu-baseline --no-secondary --no-pool --steps 1 --warmup 0 > /tmp/pmc_log_ic.txt 2> /tmp/pmc_err_ic.txt
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
f = glob.glob("/tmp/pmc_out_ic/**/*counter_collection.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    acc[r["Kernel_Name"].split("(")[0].replace("void abpoa_hip::", "")][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if k.startswith(("dp_", "poa_rounds")): print(k, {c: "%.4g" % x for c, x in sorted(v.items())})
PY
