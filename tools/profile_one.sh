# One workload of tools/profile_round.sh (kernel table + PMC traffic + PMC instruction counts).  usage: bash tools/profile_one.sh <tag> <cfg2|cfg3|cfg4|cfg5> <sets> <steps> <warmup>
R=$GRAFT_REPO_ROOT; TAG=${1:-r5}; WL=$2; N=$3; ST=$4; WU=$5
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
[ "$WL" = "cfg3" ] && export ABPOA_HIP_FIRST_PASS=1 || unset ABPOA_HIP_FIRST_PASS
rm -rf /tmp/ks_$WL
rocprofv3 --kernel-trace --stats -d /tmp/ks_$WL -o p --output-format csv -- python3 $R/bench.py --workload $WL --sets $N --steps $ST --warmup $WU --no-cpu-baseline --no-secondary --no-pool > $R/gpurun_out/${TAG}_bench_${WL}.json 2> /tmp/ks_$WL.err
cp $(find /tmp/ks_$WL -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_bench_${WL}_kernel_stats.csv
echo "== $WL kernel stats done: $(date +%T)"
(cd $R && bash tools/pmc_traffic.sh $WL $N > gpurun_out/pmc_t_$WL.log 2>&1; bash tools/pmc_insts.sh $WL $N > gpurun_out/pmc_i_$WL.log 2>&1)
echo "== $WL pmc done: $(date +%T)"
