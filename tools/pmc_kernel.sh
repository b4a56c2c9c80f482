# Any PMC counters for one kernel of a bench workload:  bash tools/pmc_kernel.sh <workload> <sets> <kernel substring> <counter> [<counter> ...]   (GPU box)
# One rocprofv3 --pmc pass per group of up to four counters; prints per-kernel totals over the run (one bench step, no warm-up).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=$1; N=$2; KERN=$3; shift 3
ARGS="--workload $WL --sets $N --no-cpu-baseline --no-secondary --no-pool --steps 1 --warmup 0"
i=0
while [ $# -gt 0 ]; do
  grp="$1 $2 $3 $4"; shift; shift 2>/dev/null; shift 2>/dev/null; shift 2>/dev/null
  rm -rf /tmp/pmc_k_$i
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_k_$i -o p --output-format csv -- python3 $R/bench.py $ARGS > /tmp/pmc_k_$i.log 2> /tmp/pmc_k_$i.err
  python3 - "$KERN" /tmp/pmc_k_$i <<'PY'
import csv, glob, sys, collections
kern, d = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); n = collections.Counter()
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if kern in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for c in sorted(acc): print(f"{kern} {c} total {acc[c]:.4g} over {n[c]} launches")
PY
  i=$((i+1))
done
