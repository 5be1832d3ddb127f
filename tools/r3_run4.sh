set -x
OUT=gpurun_out/r3d; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log; tail -4 $OUT/pytest.log
# A/B: 12-limb G1 accumulate at 3 waves per SIMD (168 VGPRs + 44 spilled) vs 2 (192 VGPRs, no scratch)
for L in libavrf.so libavrf_g2.so; do
  for rep in 1 2; do
    AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ring_bench.py 1024 2048 1 2>&1 | tail -1 | sed "s/^/$L 1ctx: /"
    AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ring_bench.py 1024 4096 4 2>&1 | tail -1 | sed "s/^/$L 4ctx: /"
  done
done
