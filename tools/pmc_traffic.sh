# HBM traffic of the DP kernels on the default bench workload: FETCH_SIZE and WRITE_SIZE in separate --pmc passes (they do not
# fit one pass; no tracing flag besides --kernel-trace).  Prints the per-launch average per kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_out -o p --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 1 --warmup 0 > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$c" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if r["Counter_Name"] != sys.argv[2]: continue
    k = r["Kernel_Name"][:48]; acc[k] += float(r["Counter_Value"]); cnt[k] += 1
for k in acc: print(f"{sys.argv[2]:10s} {k:50s} launches {cnt[k]:4d}  avg per launch {acc[k]/cnt[k]:14.1f}  total {acc[k]:16.1f}")
PY
done
