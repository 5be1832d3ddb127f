set -x
mkdir -p gpurun_out/r2a
nproc > gpurun_out/r2a/host.txt; lscpu | head -20 >> gpurun_out/r2a/host.txt; free -g >> gpurun_out/r2a/host.txt
timeout 1500 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_thin_batch.py tests/test_gpu_thin_single.py tests/test_gpu_pedersen.py -m gpu -x -q > gpurun_out/r2a/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2a/pytest.log
hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench tools/ubench.hip && timeout 300 /tmp/ubench > gpurun_out/r2a/ubench.txt 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r2a/prof_single -o single -- python3 $GRAFT_REPO_ROOT/bench.py --streams 1 --steps 20 --warmup 3 --no-ring --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r2a/bench_single.json 2> $GRAFT_REPO_ROOT/gpurun_out/r2a/bench_single.err
cd $GRAFT_REPO_ROOT
python tools/kstats.py gpurun_out/r2a/prof_single > gpurun_out/r2a/kstats_single.txt
find gpurun_out/r2a/prof_single -name "*.csv" ! -name "*kernel_stats*" -delete; find gpurun_out/r2a/prof_single -name "*.db" -delete
tail -3 gpurun_out/r2a/pytest.log; cat gpurun_out/r2a/kstats_single.txt
