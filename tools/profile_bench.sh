# rocprofv3 kernel trace + stats of the default bench command; summaries are copied to profiles/ by hand
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/prof_out
rocprofv3 --kernel-trace --stats -d /tmp/prof_out -o bench --output-format csv -- python3 $R/bench.py --no-cpu-baseline --steps 3 --warmup 1 > $R/gpurun_out/prof_bench.json 2> $R/gpurun_out/prof_bench.err
find /tmp/prof_out -name "*kernel_stats.csv" -exec cp {} $R/gpurun_out/prof_kernel_stats.csv \;
head -12 $R/gpurun_out/prof_kernel_stats.csv
cat $R/gpurun_out/prof_bench.json | cut -c1-400
