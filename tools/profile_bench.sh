# Kernel table of the default bench command: rocprofv3 --kernel-trace --stats.  Copies the per-kernel summary to gpurun_out/ (then commit it under
# profiles/).  usage (on the GPU box): bash tools/profile_bench.sh [extra bench.py arguments]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${TAG:-r2_bench}
rm -rf /tmp/prof_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_out -o p --output-format csv -- python3 $R/bench.py --no-pool --no-cpu-baseline "$@" > $R/gpurun_out/${TAG}.json 2> $R/gpurun_out/${TAG}.err
f=$(find /tmp/prof_out -name "*kernel_stats.csv" | head -1)
cp "$f" $R/gpurun_out/${TAG}_kernel_stats.csv
cat $R/gpurun_out/${TAG}_kernel_stats.csv | cut -c1-200
