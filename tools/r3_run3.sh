# round 3: secp256r1 on the GPU, host pairing with line tables (n = 1 ring verification), full suite
set -x
OUT=gpurun_out/r3c
mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_secp256r1.py -m gpu -x -q > $OUT/pytest_secp.log 2>&1; echo "rc=$?" >> $OUT/pytest_secp.log
tail -30 $OUT/pytest_secp.log
timeout 1500 python -m pytest tests -m gpu -x -q --deselect tests/test_gpu_secp256r1.py > $OUT/pytest.log 2>&1; echo "pytest rc=$?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
timeout 600 python bench.py --ring-only > $OUT/ring.json 2> $OUT/ring.err
python - <<PY
import json
d=json.loads(open("$OUT/ring.json").read().strip().splitlines()[-1])
print({k: (round(v,1) if isinstance(v,float) else v) for k,v in d.items() if not isinstance(v,dict) and k!="workload"}, d.get("ring_verify_single_ms_by_entry_point"))
print("   c4:", {k: round(v,1) for k,v in d["configs4_shape"].items() if isinstance(v,float)}, d["configs4_shape"].get("ring_verify_single_ms_by_entry_point"))
PY
AVRF_RING_TRACE=1 python tools/single_verify.py 2>&1 | tail -20
