cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_ddp.py -m gpu -x -q -s 2>&1 | tail -25 > gpurun_out/r6_rccl_tests.txt
cat gpurun_out/r6_rccl_tests.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs > gpurun_out/r6_bench_nogroup.txt 2>&1
python bench.py --rccl --steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-exact --no-full-step --no-other-configs > gpurun_out/r6_bench_rccl1.txt 2>&1
grep -h "^{" gpurun_out/r6_bench_nogroup.txt gpurun_out/r6_bench_rccl1.txt | cut -c1-600
