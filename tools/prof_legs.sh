# rocprofv3 kernel stats of bench.py's config 3 / 4 / 5 legs, one run each.  Run on the GPU box; summaries -> gpurun_out/r02_leg_*.md
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for leg in config3 config4 config5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$leg -o p -- python3 $R/tools/legs_only.py $leg > $R/gpurun_out/prof_$leg.log 2>&1 || (tail -5 $R/gpurun_out/prof_$leg.log; exit 1)
done
cd $R
python tools/profile_summary.py stats gpurun_out/prof_config3/p_kernel_stats.csv 12 gpurun_out/r02_leg_config3_vq_argmin_kernel_stats.md
python tools/profile_summary.py stats gpurun_out/prof_config4/p_kernel_stats.csv 18 gpurun_out/r02_leg_config4_vicreg_step_kernel_stats.md
python tools/profile_summary.py stats gpurun_out/prof_config5/p_kernel_stats.csv 18 gpurun_out/r02_leg_config5_ntxent_step_kernel_stats.md
head -12 gpurun_out/r02_leg_config3_vq_argmin_kernel_stats.md; head -16 gpurun_out/r02_leg_config4_vicreg_step_kernel_stats.md; tail -1 gpurun_out/r02_leg_config4_vicreg_step_kernel_stats.md; tail -1 gpurun_out/r02_leg_config5_ntxent_step_kernel_stats.md
