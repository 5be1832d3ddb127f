set -x
OUT=gpurun_out/r3e; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
for s in 0 1 3 4; do python tools/ped_bench.py 65536 $s 2>&1 | grep -v "^W2\|^E2"; done
