set -x
OUT=gpurun_out/r3f; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
# the same suite and the 25-round interleaved-suites stress (x4) on a build WITH interprocedural register allocation
export AVRF_LIB_PATH=$PWD/ark_vrf_amd/libavrf_ipra.so
for i in 1 2 3 4; do timeout 600 python -m pytest tests/test_gpu_repeatability.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/ipra stress $i: /"; done | tee $OUT/ipra_stress.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 | tee $OUT/ipra_pytest.txt
