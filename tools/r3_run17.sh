# A/B: per-item kernels of suite 0 with the whole call tree inlined under __launch_bounds__(128, 2) (256 VGPRs, 0.6-1 KB of scratch,
# two waves per SIMD) against the out-of-line tree at one wave per SIMD (256 + 105-220 registers, 1.5-2.4 KB of scratch)
for rep in 1 2; do for L in libavrf.so libavrf_w2.so; do AVRF_LIB_PATH=$PWD/ark_vrf_amd/$L python tools/ped_bench.py 2>&1 | grep "/s" | sed "s/^/$L  /"; done; done
AVRF_LIB_PATH=$PWD/ark_vrf_amd/libavrf_w2.so python -m pytest tests -m gpu -x -q -k "thin or pedersen or tiny or wire or fullsize or vectors" 2>&1 | tail -2
