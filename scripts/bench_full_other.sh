#!/bin/bash
# secondary: the full GAN step at the other stages (C4 / C5 / r128), one JSON line each
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
: > gpurun_out/${OUT:-r4_full_gan_step.jsonl}
python scripts/bench_full_step.py --steps 3 --warmup 1 >> gpurun_out/${OUT:-r4_full_gan_step.jsonl} 2>/dev/null
python scripts/bench_full_step.py --steps 3 --warmup 1 --img-size 256 --batch 4 --num-steps 24 --freeze --diffaug --no-aux >> gpurun_out/${OUT:-r4_full_gan_step.jsonl} 2>/dev/null
python scripts/bench_full_step.py --steps 3 --warmup 1 --img-size 256 --batch 4 --num-steps 12 --freeze --diffaug >> gpurun_out/${OUT:-r4_full_gan_step.jsonl} 2>/dev/null
python scripts/bench_full_step.py --steps 3 --warmup 1 --img-size 128 --batch 8 >> gpurun_out/${OUT:-r4_full_gan_step.jsonl} 2>/dev/null
cat gpurun_out/${OUT:-r4_full_gan_step.jsonl}
