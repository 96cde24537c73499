# rocprofv3 kernel stats of bench.py's config 3 / 4 / 5 legs (one run).  Run on the GPU box; summary -> gpurun_out/<tag>_legs_kernel_stats.md (usage: bash tools/prof_legs.sh [tag=r04])
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_legs -o p -- python3 $R/bench.py --legs-only --leg config3_vq_argmin,config4_vicreg_step,config5_ntxent_step --repeats 1 --no-cpu-baseline > $R/gpurun_out/prof_legs.log 2>&1 || (tail -5 $R/gpurun_out/prof_legs.log; exit 1)
cd $R
python tools/profile_summary.py stats gpurun_out/prof_legs/p_kernel_stats.csv 1 gpurun_out/${1:-r04}_legs_kernel_stats.md
head -40 gpurun_out/${1:-r04}_legs_kernel_stats.md
