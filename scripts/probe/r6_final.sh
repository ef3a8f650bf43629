cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r6_gpu_suite2.log
tail -3 gpurun_out/r6_gpu_suite2.log
python __graft_entry__.py smoke > gpurun_out/r6_smoke.log 2>&1; tail -1 gpurun_out/r6_smoke.log
python bench.py > gpurun_out/r6_bench_c2.json 2> gpurun_out/r6_bench_c2.err
tail -c 300 gpurun_out/r6_bench_c2.json
bash scripts/r6_profiles.sh stats > gpurun_out/r6_profiles.log 2>&1
OUT=r6_full_gan_step.jsonl bash scripts/bench_full_other.sh > /dev/null 2>&1
cut -c1-330 gpurun_out/r6_full_gan_step.jsonl
