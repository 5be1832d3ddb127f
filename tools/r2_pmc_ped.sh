OUT=gpurun_out/r2q; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace --pmc VALUBusy SQ_INSTS_VALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_ped -o v -- python3 $GRAFT_REPO_ROOT/tools/ped_bench.py > $GRAFT_REPO_ROOT/$OUT/pmc_ped.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_ANY --output-format csv -d $GRAFT_REPO_ROOT/$OUT/pmc_ped2 -o v -- python3 $GRAFT_REPO_ROOT/tools/ped_bench.py >> $GRAFT_REPO_ROOT/$OUT/pmc_ped.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<EOF
import csv,glob,re
from collections import defaultdict
acc=defaultdict(lambda: defaultdict(lambda:[0.0,0]))
for f in glob.glob("$OUT/pmc_ped*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        name=re.sub(r"\(.*","",r["Kernel_Name"]).replace("void ","").strip()
        a=acc[name][r["Counter_Name"]]; a[0]+=float(r["Counter_Value"]); a[1]+=1
for k,d in acc.items():
    if "ped_" in k or "thin_" in k or "k_smul" in k: print(k[:60],{c:round(v[0]/max(1,v[1]),1) for c,v in d.items()})
EOF
find $OUT -name "*.csv" -delete
