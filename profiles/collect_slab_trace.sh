#!/bin/bash
# Kernel trace of ONE rank's share of the metric's 8-GPU point (run on the GPU box from the repo root):
#   bash profiles/collect_slab_trace.sh [tag]
# C3 cut into 8 z-slabs gives every rank a 256 x 256 x 32 lattice (2,097,152 particles); this traces that slab alone
# (bench.py --force-slab --lattice 256,256,32: the slab step of ONE slab with no neighbours -- sort chain, bounds, split density, force, the
# host wait of the real step, no transfers) and summarises the timed window with profiles/step_trace_summary.py.
set -e
# kernel traces are taken with EVENT hops: the product's write / wait-value hops are spinning one-workgroup kernels of the runtime
# (__amd_rocclr_streamOpsWait), which a kernel trace counts as device-busy time and which rocprofv3's own serialisation slows down
export SPH_SLAB_HOPS=event
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_slab8; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o s8 -- python bench.py --force-slab --lattice 256,256,32 --runup 6000 --steps 200 --warmup 20 --no-cpu > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > gpurun_out/${TAG}_slab_one_eighth_bench_under_rocprof.json
python profiles/step_trace_summary.py $OUT/s8_kernel_trace.csv k_slab_bounds 6020 200 gpurun_out/${TAG}_slab_one_eighth_kernel_stats.csv > /dev/null
rm -rf $OUT
cat gpurun_out/${TAG}_slab_one_eighth_kernel_stats.csv
