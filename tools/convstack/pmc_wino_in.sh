cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_in -- python3 $R/tools/convstack/wino_in_bench.py > $R/gpurun_out/pmc_in_stdout.txt 2>&1
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
f=glob.glob(R+"/gpurun_out/pmc_in/**/*counter_collection.csv", recursive=True)[0]
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    k=r["Kernel_Name"]
    if "wino" in k or "gn_apply" in k:
        acc[k[:60]+"|"+r["Grid_Size"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: "%.3g"%(sum(x)/len(x)) for c,x in v.items()})
PY
