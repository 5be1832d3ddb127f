# A/B of per-item kernel builds of suite 0 against the shipped one (AVRF_LIB_PATH)

for rep in 1 2; do for L in libavrf.so libavrf_pc.so; do AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ped_bench.py 2>&1 | grep "/s" | sed "s/^/$L  /"; done; done
AVRF_LIB_PATH=$PWD/ark_vrf_amd/libavrf_pc.so python -m pytest tests -m gpu -x -q -k "thin or pedersen or tiny or wire or fullsize or vectors" 2>&1 | tail -2
