# PMC counters of the fast-path kernels (row loop and backtrack tail) on the kernel micro-benchmark with cigars on; per alignment
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_INSTS_BRANCH SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_out -o p --output-format csv -- python3 $R/tools/kernel_bench.py ${1:-s1k_ag_gb/aln_011} 1000 1 > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "dp_fast" in k: print(("tail " if "tail" in k else "rows ") + k[-40:], {c: round(x / 3 / 1000, 0) for c, x in v.items()}, "(per alignment)")
PY
done
tail -3 /tmp/pmc_log.txt
