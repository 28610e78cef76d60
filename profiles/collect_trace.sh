#!/bin/bash
# Kernel trace of the default bench command (run on the GPU box from the repo root):
#   bash profiles/collect_trace.sh [tag]      (tag = r04: prefix of the output files)
# rocprofv3 --kernel-trace --stats over `python bench.py --steps 100 --warmup 20 --no-cpu --no-pmc`; profiles/trace_window.py cuts the
# timed windows out of the trace (bench.py steps the dam 6000 times before it times anything):
#   steps 6021..6120  the flowing dam, merge sort      -> profiles/${TAG}_c3_flow_kernel_stats.csv
#   per-dispatch durations of k_force / k_density      -> profiles/${TAG}_c3_outlier_launches.txt (+ the two series as JSON)
#   steps 6224..6323  the same state, full radix sort   -> profiles/${TAG}_c3_fullsort_kernel_stats.csv
# and the JSON line the bench printed under the profiler -> profiles/${TAG}_c3_flow_bench_under_rocprof.json
set -e
TAG=${1:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_trace; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o c3 -- python bench.py --steps 100 --warmup 20 --no-cpu --no-pmc > $OUT/bench.log 2>&1
python - <<PY
import re
txt = open("$OUT/bench.log").read()
m = re.findall(r'^\{"metric".*\}$', txt, flags=re.M)
open("gpurun_out/" + "$TAG" + "_c3_flow_bench_under_rocprof.json", "w").write(m[-1] + "\n")
PY
python profiles/trace_window.py $OUT/c3_kernel_trace.csv 6020 100 gpurun_out/${TAG}_c3_flow_kernel_stats.csv > /dev/null
python profiles/trace_window.py $OUT/c3_kernel_trace.csv 6223 100 gpurun_out/${TAG}_c3_fullsort_kernel_stats.csv > /dev/null
cp $OUT/c3_kernel_stats.csv gpurun_out/${TAG}_c3_whole_run_kernel_stats.csv
# which steps do the slow launches belong to?  (k_force max 4.52 ms against 2.66 avg somewhere in the run)
python profiles/kernel_series.py $OUT/c3_kernel_trace.csv "k_force<true, true, true" gpurun_out/${TAG}_c3_k_force_series.json > gpurun_out/${TAG}_c3_outlier_launches.txt
python profiles/kernel_series.py $OUT/c3_kernel_trace.csv "k_density" gpurun_out/${TAG}_c3_k_density_series.json >> gpurun_out/${TAG}_c3_outlier_launches.txt
python profiles/kernel_series.py $OUT/c3_kernel_trace.csv "k_mm_move" >> gpurun_out/${TAG}_c3_outlier_launches.txt
rm -rf $OUT
tail -2 gpurun_out/${TAG}_c3_flow_kernel_stats.csv
