# per-launch durations of the kernels whose name contains $1 within the bench step (rocprofv3 --kernel-trace): gpurun_out/trace_<tag>.txt   usage: bash tools/trace_kernel.sh <substring> [tag]
set -e
R=$GRAFT_REPO_ROOT
TAG=${2:-t}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_$TAG -o p -- python3 $R/bench.py --steps 4 --warmup 2 --repeats 1 --no-side-stream --no-cpu-baseline --no-roofline --no-legs --no-options --no-sweep > $R/gpurun_out/trace_$TAG.log 2>&1 || (tail -5 $R/gpurun_out/trace_$TAG.log; exit 1)
cd $R
python3 - "$1" gpurun_out/trace_$TAG/p_kernel_trace.csv <<'PY'
import csv, sys
pat, path = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows if pat in r["Kernel_Name"]]
grid = [r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size", "?") for r in rows if pat in r["Kernel_Name"]]
n = len(sel)
print(f"{n} launches of *{pat}*; last step's:")
per = n // 6 if n % 6 == 0 else n
print(" ".join(f"{v:.0f}({g})" for v, g in zip(sel[-per:], grid[-per:])))
PY
rm -rf gpurun_out/trace_$TAG
