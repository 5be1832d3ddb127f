# A/B of library builds on the ring prover: tools/r2_ab_ring.sh <lib1> <lib2> ...   ("default" = the in-tree library)
for L in "$@"; do
  if [ "$L" = default ]; then unset AVRF_LIB_PATH; else export AVRF_LIB_PATH=$PWD/$L; fi
  echo "== $L"
  timeout 600 python tools/ring_bench.py 1024 2048 1 2>&1 | grep -E "proofs/s|verif" | head -4
  timeout 600 python tools/ring_bench.py 1024 4096 4 2>&1 | grep -E "contexts|proofs/s" | tail -2
done
