#!/bin/bash
# Round-6 numbers, part B: BASELINE config 5 (134,217,728 particles, 1024^3 cells) on ONE MI355X in both arithmetics.
set -e
python bench.py --workload C5 --steps 20 --warmup 5 --no-cpu --no-pmc 2> gpurun_out/r06_c5_f32.err | grep '^{"metric"' > gpurun_out/r06_c5_bench_f32.json
echo "C5 f32 done"
python bench.py --workload C5 --precision mixed --steps 20 --warmup 5 --no-cpu --no-pmc 2> gpurun_out/r06_c5_mixed.err | grep '^{"metric"' > gpurun_out/r06_c5_bench_mixed.json
echo "C5 mixed done"
