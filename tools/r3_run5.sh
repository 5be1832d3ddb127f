set -x
for L in libavrf.so libavrf_g2.so; do
  for rep in 1 2; do
    AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ring_bench.py 1024 2048 1 2>&1 | tail -1 | sed "s/^/$L 1ctx: /"
    AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ring_bench.py 1024 4096 4 2>&1 | tail -1 | sed "s/^/$L 4ctx: /"
  done
done
