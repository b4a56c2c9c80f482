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
# idle time on the GPU between consecutive kernels of the step (launch gaps, host stalls), and time per kernel family
allk = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
gap = sum(max(0, allk[i + 1][1] - max(e for _, _, e in allk[:i + 1][-4:])) for i in range(len(allk) - 1)) / 1e6
span = (max(e for _, _, e in allk) - allk[0][1]) / 1e6
fam = {}
for k, s_, e in allk:
    n = k.split("abpoa_hip::")[-1].split("(")[0].split("<")[0]; fam[n] = fam.get(n, 0) + (e - s_) / 1e6
print(f"GPU span {span:.1f} ms, idle between kernels {gap:.1f} ms, by kernel (ms): " + ", ".join(f"{k} {v:.1f}" for k, v in sorted(fam.items(), key=lambda x: -x[1])))
# print compactly: one line per round = consecutive kernels until the next first-row kernel
line = []
for k, d in out:
    tag = k.replace("dp_wide_kernel", "W").replace("dp_fast_tail_kernel", "T").replace("dp_fast_kernel", "F")
    line.append(f"{tag}={d:.2f}")
for i in range(0, len(line), 4): print("  ".join(line[i:i + 4]))
PY
