cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/r6_bench_c2.json 2> gpurun_out/r6_bench_c2.err
tail -c 600 gpurun_out/r6_bench_c2.json
bash scripts/r6_profiles.sh all > gpurun_out/r6_profiles.log 2>&1
OUT=r6_full_gan_step.jsonl bash scripts/bench_full_other.sh > /dev/null 2>&1
cat gpurun_out/r6_full_gan_step.jsonl | cut -c1-400
