#!/bin/bash
# Round-6 numbers, part A (GPU box, repo root): the default bench line, its kernel trace and counter passes, and the small
# configurations WITHOUT the profiler.     bash profiles/collect_r06_a.sh
set -e
python bench.py > gpurun_out/r06_bench_default_run.log 2> gpurun_out/r06_bench_default_run.err
grep '^{"metric"' gpurun_out/r06_bench_default_run.log > gpurun_out/r06_bench_default_run.json
echo "default bench done"
bash profiles/collect_trace.sh r06 > gpurun_out/r06_collect_trace.log 2>&1
echo "trace done"
bash profiles/collect_pmc.sh 6000 20 5 r06 > gpurun_out/r06_collect_pmc.log 2>&1
echo "pmc done"
python bench.py --workload C2 --steps 200 --warmup 20 > gpurun_out/r06_c2_bench.log 2>/dev/null
grep '^{"metric"' gpurun_out/r06_c2_bench.log > gpurun_out/r06_c2_bench.json
echo "C2 done"
: > gpurun_out/r06_headless_sizes.txt
for N in 128 8192 131072 262144; do
  echo "== sph_headless -benchmark -n=$N -i=2000 (default box 2, 32^3 cells, as upstream)" >> gpurun_out/r06_headless_sizes.txt
  gpufluidsimulator_amd/sph_headless -benchmark -n=$N -i=2000 | tail -2 >> gpurun_out/r06_headless_sizes.txt
done
gpufluidsimulator_amd/sph_headless -benchmark -n=131072 -i=4000 -log=gpurun_out/r06_headless_n131072_oscar_log.txt -logfreq=100 -logstyle=oscar > /dev/null
python tools/bench_log_summary.py gpurun_out/r06_headless_n131072_oscar_log.txt tests/golden/oscar_logs/n131072_CUDA.txt > gpurun_out/r06_log_summary_old_and_new.txt
echo "headless done"
python bench.py --force-slab --lattice 256,256,32 --steps 200 --warmup 20 --no-cpu --no-pmc 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_slab_one_eighth_bench.json
python bench.py --force-slab --periodic-z --link-latency-us 10 --link-gbs 153 --early-force off --steps 200 --warmup 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_periodic_slab_one_eighth_153gbs_10us.json
python bench.py --force-slab --periodic-z --link-latency-us 0 --link-gbs 0 --early-force off --steps 200 --warmup 20 2>/dev/null | grep '^{"metric"' > gpurun_out/r06_periodic_slab_one_eighth_no_link_delay.json
echo "slab done"
