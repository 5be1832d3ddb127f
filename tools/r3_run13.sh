# A/B: 12-limb G1 reduction kernels (k_wsum, k_wsum_blk) at 2 waves per SIMD (256 VGPRs + 160 / 284 bytes of scratch) against 1 (295 / 303 registers)
for rep in 1 2 3; do
  for L in libavrf.so libavrf_r2.so; do
    AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ring_bench.py 1024 4096 4 2>&1 | grep "proofs/s" | sed "s/^/$L /"
  done
done
