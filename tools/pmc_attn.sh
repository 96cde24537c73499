set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU --output-format csv -d $R/gpurun_out/pmc_attn -o p -- python3 $R/tools/attn_ab.py 256 3 1 > $R/gpurun_out/pmc_attn.log 2>&1 || (tail -5 $R/gpurun_out/pmc_attn.log; exit 1)
cd $R
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_attn/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "attn" in r["Kernel_Name"]:
            acc[r["Kernel_Name"].split("(")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,cs in acc.items():
    d={c:sum(v)/len(v) for c,v in cs.items()}
    busy=d.get("SQ_BUSY_CU_CYCLES",1)
    print(k, {c: round(v/ (4*busy) if c!="SQ_INSTS_VALU" else v,3) for c,v in d.items() if c not in ("SQ_BUSY_CU_CYCLES",)})
PY
