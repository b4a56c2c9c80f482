# One GPU-box call that produces every profile record of a round (copied to profiles/ afterwards):
#   kernel tables (rocprofv3 --kernel-trace --stats) of the four bench workloads, PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and PMC instruction
#   counts for each.  usage: bash tools/profile_round.sh <round tag, e.g. r4>
R=$GRAFT_REPO_ROOT; TAG=${1:-r5}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for spec in "cfg2 1000 5 2" "cfg5 1000 3 1" "cfg4 2048 2 1" "cfg3 2048 1 1"; do
  set -- $spec; WL=$1; N=$2; ST=$3; WU=$4
  [ "$WL" = "cfg3" ] && export ABPOA_HIP_FIRST_PASS=1 || unset ABPOA_HIP_FIRST_PASS
  rm -rf /tmp/ks_$WL
  rocprofv3 --kernel-trace --stats -d /tmp/ks_$WL -o p --output-format csv -- python3 $R/bench.py --workload $WL --sets $N --steps $ST --warmup $WU --no-cpu-baseline --no-secondary --no-pool > $R/gpurun_out/${TAG}_bench_${WL}.json 2> /tmp/ks_$WL.err
  cp $(find /tmp/ks_$WL -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_${WL}_kernel_stats.csv
  echo "== $WL kernel stats done: $(date +%T)"
  (cd $R && bash tools/pmc_traffic.sh $WL $N > gpurun_out/pmc_t_$WL.log 2>&1; bash tools/pmc_insts.sh $WL $N > gpurun_out/pmc_i_$WL.log 2>&1)
  echo "== $WL pmc done: $(date +%T)"
done
