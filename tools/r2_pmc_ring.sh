OUT=gpurun_out/r2q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc VALUBusy VALUUtilization --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_valu -o v -- python3 $GRAFT_REPO_ROOT/tools/ring_bench.py 1024 512 1 > $GRAFT_REPO_ROOT/$OUT/pmc.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<EOF
import csv,glob,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/pmc_valu/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").strip()
        a=acc[name][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,d in acc.items():
    if "accumulate" in k or "wsum" in k or "bucket" in k or "ntt" in k: print(k[:80],{c:round(v[0]/max(1,v[1]),1) for c,v in d.items()}, list(d.values())[0][1])
EOF
find $OUT -name "*.csv" -delete
