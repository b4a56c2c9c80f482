cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for grp in "SQ_IFETCH SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_LDS" "SQ_INST_LEVEL_LDS SQ_INSTS_LDS_LOAD" "SQ_INST_CYCLES_VMEM_WR SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM" "SQ_IFETCH_LEVEL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_CMD_FIFO_FULL"; do
  rm -rf /tmp/pmc_out
  rocprofv3 --kernel-trace --pmc $grp -d /tmp/pmc_out -o p --output-format csv -- python3 $R/tools/kernel_bench.py ${1:-s1k_ag_gb/aln_011} 1000 ${2:-0} > /tmp/pmc_log.txt 2>&1
  f=$(find /tmp/pmc_out -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"][:40]; acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    if "dp_fast_kernel" in k: print({c: round(x / 3 / 1357e3, 1) for c, x in v.items()}, "(per row)")
PY
done
