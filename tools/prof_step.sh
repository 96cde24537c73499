# rocprofv3 kernel stats of the B = 1024 step (weight gradients on the main stream): run on the GPU box, summary -> gpurun_out/r03_kernel_stats.md
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03 -o p -- python3 $R/bench.py --steps 15 --warmup 3 --repeats 1 --no-side-stream --no-cpu-baseline --no-roofline --no-legs --no-options > $R/gpurun_out/prof_r03.log 2>&1 || (tail -5 $R/gpurun_out/prof_r03.log; exit 1)
cd $R
ls gpurun_out/prof_r03 | head
python tools/profile_summary.py stats $(ls gpurun_out/prof_r03/*kernel_stats.csv | head -1) 18 gpurun_out/r03_kernel_stats.md   # 3 warm-up + 15 timed steps were profiled
head -24 gpurun_out/r03_kernel_stats.md
