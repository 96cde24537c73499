# same-box A/B of two library builds on the whole step: bash tools/ab_step.sh <base.so> [rounds]  -> gpurun_out/ab_step.txt
# (new = the in-tree library; runs base, new, base, new ... one process after the other; each prints ms/step of 3 x 30 steps)
R=${GRAFT_REPO_ROOT:-.}
BASE=$1; N=${2:-2}
OUT=$R/gpurun_out/ab_step.txt; : > $OUT
FL="--steps 30 --warmup 10 --repeats 3 --no-cpu-baseline --no-roofline --no-legs --no-options"
for i in $(seq $N); do
  python3 $R/tools/lib_ab.py $BASE $R/bench.py $FL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('base', d['ms_per_step'], d['protocol']['repeats_ms_per_step'])" >> $OUT || exit 1
  python3 $R/bench.py $FL 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('new ', d['ms_per_step'], d['protocol']['repeats_ms_per_step'])" >> $OUT || exit 1
done
cat $OUT
