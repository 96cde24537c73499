# HBM traffic of the legs' dominant kernels: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace domains) of
# `bench.py --legs-only --leg <name>`, one leg per run.  Run on the GPU box; summary -> gpurun_out/<tag>_pmc_legs_traffic.json
# usage: bash tools/pmc_legs.sh [tag=r04]
set -e
R=$GRAFT_REPO_ROOT
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp
for leg in config3_vq_argmin config4_vicreg_step config5_ntxent_step; do
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmcl_${leg}_$c -o p -- python3 $R/bench.py --legs-only --leg $leg --repeats 1 --no-cpu-baseline > $R/gpurun_out/pmcl_${leg}_$c.log 2>&1 || (tail -5 $R/gpurun_out/pmcl_${leg}_$c.log; exit 1)
    echo "$leg $c done"
  done
done
cd $R
python tools/profile_summary.py pmc_legs gpurun_out gpurun_out/${TAG}_pmc_legs_traffic.json
rm -rf gpurun_out/pmcl_*
cat gpurun_out/${TAG}_pmc_legs_traffic.json | head -40
