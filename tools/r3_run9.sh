set -x
OUT=gpurun_out/r3g; mkdir -p $OUT
# the 25-round interleaved-suites stress (x6) and the whole suite on a build WITH interprocedural register allocation (make IPRA=1)
export AVRF_LIB_PATH=$PWD/ark_vrf_amd/libavrf_ipra.so
for i in 1 2 3 4 5 6; do timeout 600 python -m pytest tests/test_gpu_repeatability.py -m gpu -x -q 2>&1 | tail -1 | sed "s/^/ipra stress $i: /"; done | tee $OUT/ipra_stress.txt
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | tee $OUT/ipra_pytest.txt
