#!/bin/bash
# Kernel trace of ONE slab between its periodic images (a middle rank's whole step), early force off and on:
#   bash profiles/collect_periodic_trace.sh [tag] [latency_us] [gbs]
set -e
# kernel traces are taken with EVENT hops: the product's write / wait-value hops are spinning one-workgroup kernels of the runtime
# (__amd_rocclr_streamOpsWait), which a kernel trace counts as device-busy time and which rocprofv3's own serialisation slows down
export SPH_SLAB_HOPS=event
TAG=${1:-r05}; LAT=${2:-0}; GBS=${3:-0}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for EF in off on; do
  OUT=gpurun_out/prof_periodic_$EF; rm -rf $OUT; mkdir -p $OUT
  rocprofv3 --kernel-trace --output-format csv -d $OUT -o p -- python bench.py --force-slab --periodic-z --link-latency-us $LAT --link-gbs $GBS --early-force $EF --runup 6000 --steps 200 --warmup 20 > $OUT/bench.log 2>&1
  python profiles/step_trace_summary.py $OUT/p_kernel_trace.csv k_slab_bounds 6020 200 gpurun_out/${TAG}_periodic_slab_early_${EF}_kernel_stats.csv > /dev/null
  rm -rf $OUT
  echo "== early force $EF =="; cat gpurun_out/${TAG}_periodic_slab_early_${EF}_kernel_stats.csv
done
