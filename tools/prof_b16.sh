set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b16 -o p -- python3 $R/bench.py --batch 16 --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-roofline --no-legs --no-options --no-sweep > $R/gpurun_out/prof_b16.log 2>&1 || (tail -5 $R/gpurun_out/prof_b16.log; exit 1)
cd $R
python tools/profile_summary.py stats $(ls gpurun_out/prof_b16/*kernel_stats.csv | head -1) 35 gpurun_out/r04_b16_kernel_stats.md
rm -rf gpurun_out/prof_b16
head -36 gpurun_out/r04_b16_kernel_stats.md; tail -2 gpurun_out/r04_b16_kernel_stats.md; grep -o '"ms_per_step": [0-9.]*' gpurun_out/prof_b16.log | head -1
