#!/bin/bash
# Two ranks of the slab step in ONE process on one GPU, device-to-device transport (run on the GPU box from the repo root):
# kernel trace of 50 flowing steps of two 256 x 256 x 64 slabs -> profiles/slab_overlap.py
set -e
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_overlap; rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT -o o -- python bench.py --gpus 2 --one-gpu --transport local --lattice 256,256,128 --runup 3000 --steps 50 --warmup 10 --no-cpu > $OUT/bench.log 2>&1
grep '^{"metric"' $OUT/bench.log > gpurun_out/${TAG}_slab_two_ranks_local_bench_under_rocprof.json
python profiles/slab_overlap.py $OUT/o_kernel_trace.csv 3065 2 gpurun_out/${TAG}_slab_two_ranks_local_overlap.json
rm -rf $OUT
