# rocprofv3 kernel stats of the bench step (default 2048 lines per GPU): run on the GPU box, summary -> gpurun_out/r04_kernel_stats.md (+ .csv)
# usage: bash tools/prof_step.sh [tag=r04]
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -o p -- python3 $R/bench.py --steps 15 --warmup 3 --repeats 1 --no-side-stream --no-cpu-baseline --no-roofline --no-legs --no-options --no-sweep > $R/gpurun_out/prof_$TAG.log 2>&1 || (tail -5 $R/gpurun_out/prof_$TAG.log; exit 1)
cd $R
python tools/profile_summary.py stats $(ls gpurun_out/prof_$TAG/*kernel_stats.csv | head -1) 18 gpurun_out/${TAG}_kernel_stats.md   # 3 warm-up + 15 timed steps were profiled
cp $(ls gpurun_out/prof_$TAG/*kernel_stats.csv | head -1) gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/prof_$TAG
head -30 gpurun_out/${TAG}_kernel_stats.md
