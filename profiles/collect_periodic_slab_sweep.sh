#!/bin/bash
# One slab between its periodic images (bench.py --force-slab --periodic-z: a middle rank's whole step, halo work included,
# 2,097,152 particles = an eighth of C3) for a range of emulated link latencies and bandwidths, with the early force launch
# off and on (run on the GPU box, repo root):
#   bash profiles/collect_periodic_slab_sweep.sh [tag]     -> gpurun_out/${TAG}_periodic_slab_link_sweep.txt
# The link is a PARAMETER of the loop transport (every message is held back by latency + bytes / bandwidth on the comm
# stream), not a measurement: the table says how much of a given link the step hides.
TAG=${1:-r05}
OUT=gpurun_out/${TAG}_periodic_slab_link_sweep.txt
echo "latency_us link_gbs early_force | ms_per_step(window of 200) sustained(last 1000 run-up steps) | exchange_us migrants halo_a halo_b (event pairs, mean) | host_wait_us host_pre host_post" > $OUT
for SPEC in "0 0" "10 153" "20 153" "30 153" "40 153" "80 153" "10 75" "10 40" "40 50"; do
  set -- $SPEC
  for EF in off on; do
  python bench.py --force-slab --periodic-z --link-latency-us $1 --link-gbs $2 --early-force $EF --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exchange_us']; h=d['host_step_us']
print('%5s %5s %3s | %.4f %.4f | %.1f %.1f %.1f | %.1f %.1f %.1f' % ('$1','$2','$EF',d['ms_per_step'],d['ms_per_step_sustained'],e['migrants']['mean'],e['halo_a']['mean'],e['halo_b']['mean'],d['host_wait_us']['mean'],h['host_pre_us']['mean'],h['host_post_us']['mean']))" >> $OUT
  done
done
cat $OUT
