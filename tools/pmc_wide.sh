# Instruction mix of the wide-band row loop per DP row: rocprofv3 PMC passes over the kernel micro-benchmark (one pass per counter group, no tracing
# flag besides --kernel-trace).  usage: tools/pmc_wide.sh [golden case] [copies] ; rows per alignment are read from the golden.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CASE=${1:-s10k_ag_i32/aln_008}; N=${2:-256}
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_out -o p --output-format csv -- python3 $R/tools/kernel_bench.py $CASE $N 0 > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$N" "$R" "$CASE" <<'PY'
import csv, sys, collections
sys.path.insert(0, sys.argv[3]); sys.path.insert(0, sys.argv[3] + "/tests")
import helpers as H
g = H.read_abpg([p for l, p in H.golden_cases() if l == sys.argv[4]][0]); rows = int(g["n_rows"][0])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:48]][r["Counter_Name"]] += float(r["Counter_Value"])
n = int(sys.argv[2])
for k, v in acc.items():
    if "dp_wide" in k or "dp_fast_kernel" in k or "dp_team" in k:
        if sum(v.values()) > 0: print(k, {c: round(x / 3 / n / rows, 1) for c, x in v.items()}, f"(per row; {rows} rows, 3 iterations x {n} alignments)")
PY
done
