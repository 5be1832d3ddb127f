OUT=${1:-gpurun_out/r2b}
mkdir -p $OUT
timeout 1500 python -m pytest tests/test_gpu_msm.py tests/test_gpu_g1_msm.py tests/test_gpu_thin_batch.py tests/test_gpu_fullsize.py tests/test_gpu_pedersen.py tests/test_gpu_ring.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -15 $OUT/pytest.log
bash tools/r2_prof_single.sh $OUT
cat $OUT/bench_single.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['step_breakdown_us'], d['roofline']['kernel_avg_ms'], d['config']['msm_plan'])"
