# regenerates every round-3 summary under profiles/ from the current build (run on the GPU box; the caller copies
# gpurun_out/r3fin/* to profiles/r3_*)
set -x
F=gpurun_out/r3fin; mkdir -p $F
bash tools/r2_prof_single.sh $F > /dev/null 2>&1
BENCH_ARGS="--gpus 1 --steps 20 --warmup 5" bash tools/r2_prof_multi.sh $F > $F/multi_summary.txt 2>&1
bash tools/r2_pmc.sh gpurun_out/r3fin_pmc > $F/pmc_summary.txt 2>&1; cp gpurun_out/r3fin_pmc/r2_pmc_traffic.json $F/r3_pmc_traffic.json; cp gpurun_out/r3fin_pmc/r2_pmc_valu.json $F/r3_pmc_valu.json
bash tools/r2_ring_prof.sh $F > $F/ring_summary.txt 2>&1
bash tools/r3_pmc_ring.sh gpurun_out/r3fin_pmcr > $F/pmc_ring_summary.txt 2>&1; cp gpurun_out/r3fin_pmcr/r3_pmc_*.json $F/
bash tools/r2_prof_ped.sh $F > $F/ped_summary.txt 2>&1
bash tools/r2_prof_rverify.sh $F > $F/rverify_summary.txt 2>&1
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-pass-failed -I ark_vrf_amd/csrc -o /tmp/ubench tools/ubench.hip && /tmp/ubench > $F/ubench.txt 2>&1
find $F -name "*.db" -delete
ls -la $F
