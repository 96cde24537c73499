# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) of the attention kernels: two launches vs the paired launch.  Run on the GPU box.
set -e
cd $GRAFT_REPO_ROOT
true
true
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for m in 0 1; do
 for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/attn_pmc_${m}_$c -o p -- python3 $R/tools/attn_ab.py 256 3 $m > $R/gpurun_out/attn_pmc_${m}_$c.log 2>&1
 done
done
cd $R
python - <<'PY'
import csv, glob, collections
for m in (0,1):
    for c in ("FETCH_SIZE","WRITE_SIZE"):
        acc=collections.defaultdict(list)
        for f in glob.glob(f"gpurun_out/attn_pmc_{m}_{c}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "attn" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(m, c, k, len(v), sum(v)/len(v))
PY
