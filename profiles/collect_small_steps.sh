#!/bin/bash
# Kernel traces of the SMALL steps, where the fixed (non-pair) part of a step is a large share of it (run on the GPU box
# from the repo root):    bash profiles/collect_small_steps.sh [tag]
#   (a) BASELINE config 2 (262,144 particles, 128^3 cells), flowing, whole-domain context
#   (b) the reference's own benchmark command at its largest published size: sph_headless -benchmark -n=131072 -i=2000
#   (c) one rank's eighth of C3 alone (no neighbours)             -- profiles/collect_slab_trace.sh
#   (d) the same slab between its periodic images, 10 us + 153 GB/s per message -- profiles/collect_periodic_trace.sh
# Each is summarised per step by profiles/step_trace_summary.py (kernel times, dispatches per step, idle gaps).
set -e
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_small; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o c2 -- python bench.py --workload C2 --steps 200 --warmup 20 --no-cpu --no-pmc > $OUT/c2.log 2>&1
grep '^{"metric"' $OUT/c2.log > gpurun_out/${TAG}_c2_bench_under_rocprof.json
python profiles/step_trace_summary.py $OUT/c2_kernel_trace.csv k_force 6020 200 gpurun_out/${TAG}_c2_flow_kernel_stats.csv > /dev/null
rm -f $OUT/c2_kernel_trace.csv
echo "== (a) C2 flowing =="; cat gpurun_out/${TAG}_c2_flow_kernel_stats.csv
rocprofv3 --kernel-trace --output-format csv -d $OUT -o h -- gpufluidsimulator_amd/sph_headless -benchmark -n=131072 -i=2000 > $OUT/h.log 2>&1
tail -2 $OUT/h.log > gpurun_out/${TAG}_headless_n131072.txt
python profiles/step_trace_summary.py $OUT/h_kernel_trace.csv k_force 1000 900 gpurun_out/${TAG}_headless_n131072_kernel_stats.csv > /dev/null
rm -rf $OUT
echo "== (b) sph_headless -benchmark -n=131072 -i=2000 =="; cat gpurun_out/${TAG}_headless_n131072.txt gpurun_out/${TAG}_headless_n131072_kernel_stats.csv
echo "== (c) one eighth of C3, no neighbours =="; bash profiles/collect_slab_trace.sh $TAG
echo "== (d) periodic images, 10 us + 153 GB/s =="; bash profiles/collect_periodic_trace.sh $TAG 10 153
