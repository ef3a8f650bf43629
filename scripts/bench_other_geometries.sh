B="--steps 10 --warmup 3 --no-cpu-baseline --no-roofline --no-exact --no-full-step"
: > gpurun_out/r3_bench_other_geometries.jsonl
python bench.py $B --hier >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python bench.py $B --no-graph >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python bench.py $B --img-size 128 --batch 8 >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python bench.py $B --img-size 256 --batch 4 --hier --num-steps 24 >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python bench.py $B --img-size 256 --batch 4 --hier --num-steps 24 --freeze >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python bench.py $B --img-size 32 --batch 4 --hier --num-steps 12 >> gpurun_out/r3_bench_other_geometries.jsonl 2>/dev/null
python -c "
import json
for l in open('gpurun_out/r3_bench_other_geometries.jsonl'):
    d=json.loads(l); print(d['config'].get('workload','')[:90], d['value'], d['ms_per_step'])"
