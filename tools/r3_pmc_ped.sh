# PMC traffic of the per-item kernels (one context, 65 536 items): FETCH_SIZE and WRITE_SIZE in separate passes
OUT=${1:-gpurun_out/r3pp}; mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/tools/ped_bench.py"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_valu -o v -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_valu.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r3_pmc_traffic_per_item.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/ped_bench.py (65 536 items, one context)"
python - <<PY
import csv,glob,json,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/pmc_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").strip()
        a=acc[name][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
d=json.load(open("$OUT/r3_pmc_traffic_per_item.json"))
d["valu"]={k:{c:round(v[0]/max(1,v[1]),2) for c,v in dd.items()} for k,dd in acc.items() if "ped_" in k or "thin_" in k}
json.dump(d,open("$OUT/r3_pmc_traffic_per_item.json","w"),indent=1)
for k,v in d["kernels"].items():
    if "ped_" in k or "thin_" in k: print(k[:60].ljust(60), v, d["valu"].get(k))
PY
find $OUT -name "*.csv" -delete
