set -x
for L in libavrf_base.so libavrf_A.so libavrf_B.so; do
  echo "=== $L"
  AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ped_bench.py 65536 0 2>&1 | grep -v "^suite"
  AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L timeout 600 python -m pytest tests/test_gpu_thin_single.py tests/test_gpu_pedersen.py tests/test_gpu_tiny.py -m gpu -x -q -k "not validation and not suite" 2>&1 | tail -2
done
