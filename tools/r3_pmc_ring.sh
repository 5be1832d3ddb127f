# PMC passes of the ring prover (one context, 512 proofs = one chunk): FETCH_SIZE, WRITE_SIZE, VALU -- separate runs, folded into json
OUT=${1:-gpurun_out/r3pr}; mkdir -p $OUT
CMD="python3 $GRAFT_REPO_ROOT/tools/ring_bench.py 1024 512 1"
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_fetch -o f -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_write -o w -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_valu -o v -- $CMD > $GRAFT_REPO_ROOT/$OUT/pmc_valu.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_traffic.py $OUT/pmc_fetch $OUT/pmc_write $OUT/r3_pmc_traffic_ring.json "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 tools/ring_bench.py 1024 512 1"
python - <<PY
import csv,glob,json,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/pmc_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").strip()
        a=acc[name][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
out={"command":"rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization -- python3 tools/ring_bench.py 1024 512 1","kernels":{k:{c:round(v[0]/max(1,v[1]),2) for c,v in d.items()} for k,d in acc.items()}}
json.dump(out,open("$OUT/r3_pmc_valu_ring.json","w"),indent=1)
for k,d in out["kernels"].items():
    if "G1" in k or "ring" in k or "ntt" in k: print(k[:80],d)
PY
find $OUT -name "*.csv" -delete
