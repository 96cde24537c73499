# SQ counters of the bench step (2048 lines)'s kernels (one pass; rocprofv3 --pmc only, no trace domains).  Run on the GPU box; summary -> gpurun_out/${1:-r04}_pmc_step_sq_counters.json
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT -- python3 bench.py --steps 4 --warmup 3 --repeats 1 --no-cpu-baseline --no-options --no-legs --no-roofline --no-side-stream --no-sweep"
timeout -k 10 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT --output-format csv -d $R/gpurun_out/pmc_sq -o p -- python3 $R/bench.py --steps 4 --warmup 3 --repeats 1 --no-cpu-baseline --no-options --no-legs --no-roofline --no-side-stream --no-sweep > $R/gpurun_out/pmc_sq.log 2>&1 || (tail -5 $R/gpurun_out/pmc_sq.log; exit 1)
cd $R
python tools/pmc_summary.py gpurun_out/${1:-r04}_pmc_step_sq_counters.json "$CMD" "" $(ls gpurun_out/pmc_sq/*counter_collection.csv)
