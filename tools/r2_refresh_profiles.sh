# regenerates every summary under profiles/ from the current build (run on the GPU box; copies happen in the caller's checkout)
set -x
bash tools/r2_prof_single.sh gpurun_out/fin > /dev/null 2>&1
BENCH_ARGS="--steps 20 --warmup 5" bash tools/r2_prof_multi.sh gpurun_out/fin > gpurun_out/fin/multi_summary.txt 2>&1
bash tools/r2_pmc.sh gpurun_out/fin_pmc > gpurun_out/fin/pmc_summary.txt 2>&1; cp gpurun_out/fin_pmc/r2_pmc_*.json gpurun_out/fin/
bash tools/r2_ring_prof.sh gpurun_out/fin > gpurun_out/fin/ring_summary.txt 2>&1
bash tools/r2_prof_ped.sh gpurun_out/fin > gpurun_out/fin/ped_summary.txt 2>&1
bash tools/r2_prof_rverify.sh gpurun_out/fin > gpurun_out/fin/rverify_summary.txt 2>&1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed -I ark_vrf_amd/csrc -o /tmp/ubench tools/ubench.hip && /tmp/ubench > gpurun_out/fin/ubench.txt 2>&1
ls -la gpurun_out/fin
