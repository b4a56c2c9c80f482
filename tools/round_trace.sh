# Per-launch durations of the row-loop / tail kernels of one bench step (rocprofv3 --kernel-trace), in launch order: which rounds run both score widths.
# usage (GPU box): bash tools/round_trace.sh cfg4
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-cfg4}
rm -rf /tmp/rt_out
rocprofv3 --kernel-trace -d /tmp/rt_out -o p --output-format csv -- python3 $R/bench.py --workload $WL --no-cpu-baseline --no-secondary --no-pool --steps 1 --warmup 0 > /tmp/rt_log.txt 2> /tmp/rt_err.txt
f=$(find /tmp/rt_out -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
out = []
for r in rows:
    k = r["Kernel_Name"]
    if "dp_" not in k: continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    out.append((k.split("abpoa_hip::")[1].split("(")[0], d))
# print compactly: one line per round = consecutive kernels until the next first-row kernel
line = []
for k, d in out:
    tag = k.replace("dp_wide_kernel", "W").replace("dp_fast_tail_kernel", "T").replace("dp_fast_kernel", "F")
    line.append(f"{tag}={d:.2f}")
for i in range(0, len(line), 4): print("  ".join(line[i:i + 4]))
PY
