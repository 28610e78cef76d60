#!/bin/bash
# Per-step series of ONE rank's share of the metric's 8-GPU point (run on the GPU box from the repo root):
#   bash profiles/collect_slab_series.sh [tag] [extra bench.py flags]
# Same command as collect_slab_trace.sh (bench.py --force-slab --lattice 256,256,32: the slab step with no neighbours),
# but the WHOLE trace is kept long enough to compare the 1000 "sustained" steps (5000..5999) with the 200-step timed
# window (6020..6219) step by step (profiles/step_series.py), with rocm-smi power / clock samples beside it.
set -e
TAG=${1:-r04}; shift || true
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_slab_series; rm -rf $OUT; mkdir -p $OUT
( while true; do echo "t $(date +%s.%N)"; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk" ; sleep 0.5; done ) > $OUT/smi.log 2>&1 &
SMI=$!
rocprofv3 --kernel-trace --output-format csv -d $OUT -o s8 -- python bench.py --force-slab --lattice 256,256,32 --runup 6000 --steps 200 --warmup 20 --no-cpu "$@" > $OUT/bench.log 2>&1 || { kill $SMI; tail -20 $OUT/bench.log; exit 1; }
kill $SMI || true
grep '^{"metric"' $OUT/bench.log > gpurun_out/${TAG}_slab_series_bench.json
python profiles/step_series.py $OUT/s8_kernel_trace.csv k_slab_bounds_pack gpurun_out/${TAG}_slab_series_steps.json 4000:5000 5000:6000 6020:6220 > gpurun_out/${TAG}_slab_series_summary.txt
python profiles/step_trace_summary.py $OUT/s8_kernel_trace.csv k_slab_bounds_pack 6020 200 gpurun_out/${TAG}_slab_one_eighth_kernel_stats.csv > /dev/null
python profiles/step_trace_summary.py $OUT/s8_kernel_trace.csv k_slab_bounds_pack 5000 1000 gpurun_out/${TAG}_slab_one_eighth_kernel_stats_sustained.csv > /dev/null
cp $OUT/smi.log gpurun_out/${TAG}_slab_series_smi.log
gzip -c $OUT/s8_kernel_trace.csv > gpurun_out/${TAG}_slab_series_trace.csv.gz || true
rm -rf $OUT
cat gpurun_out/${TAG}_slab_series_summary.txt
