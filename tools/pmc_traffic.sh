# HBM traffic of the step's GEMM launches: FETCH_SIZE and WRITE_SIZE in separate rocprofv3 --pmc passes (no trace domains).
# Run on the GPU box; summary -> gpurun_out/<tag>_pmc_gemm_traffic.json (usage: bash tools/pmc_traffic.sh [tag=r04])
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 500 rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -o p -- python3 $R/bench.py --steps 3 --warmup 2 --repeats 1 --no-cpu-baseline --no-options --no-legs --no-roofline --no-side-stream --no-sweep > $R/gpurun_out/pmc_$c.log 2>&1 || (tail -5 $R/gpurun_out/pmc_$c.log; exit 1)
done
cd $R
python tools/profile_summary.py pmc gpurun_out/pmc_FETCH_SIZE/p_counter_collection.csv gpurun_out/pmc_WRITE_SIZE/p_counter_collection.csv 2048 gpurun_out/${1:-r04}_pmc_gemm_traffic.json
python -c "
import json; d=json.load(open('gpurun_out/${1:-r04}_pmc_gemm_traffic.json')); print(d['hbm_bytes_per_launch']); [print(k[:60], {a: round(b/1e6,1) if isinstance(b,float) else b for a,b in v.items()}) for k,v in d['by_kernel'].items()]"
