#!/bin/bash
# The two message protocols of the slab step on ONE slab between its periodic images (bench.py --force-slab --periodic-z: a
# middle rank's whole step, 2,097,152 particles = an eighth of C3), for a range of emulated per-message latencies at 153 GB/s
# (one xGMI link's figure), early force launch off and on (run on the GPU box, repo root):
#   bash profiles/collect_periodic_protocols.sh [tag]      -> gpurun_out/${TAG}_periodic_slab_protocols.txt
# protocol 3 = MIGRANTS / HALO A / HALO B; protocol 1 = one message (two ghost layers, ghost densities recomputed locally).
# The link is a PARAMETER of the loop transport, not a measurement.
TAG=${1:-r06}
OUT=gpurun_out/${TAG}_periodic_slab_protocols.txt
echo "protocol latency_us link_gbs early_force | ms_per_step(window of 200) sustained(last 1000 run-up steps) | exchange_us migrants halo_a halo_b one one_rest (event pairs, mean) | dens force (ms, device, summed over launches) | host_wait_us | exchanges one_steps one_rests" > $OUT
for SPEC in "0 0" "10 153" "20 153" "40 153" "80 153" "10 75"; do
  set -- $SPEC
  for P in 3 1; do
  for EF in off on; do
  python bench.py --force-slab --periodic-z --protocol $P --link-latency-us $1 --link-gbs $2 --early-force $EF --steps 200 --warmup 20 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['exchange_us']; pr=d['protocol']
f=lambda g: ('%.1f' % e[g]['mean']) if e[g]['mean'] is not None else '-'
print('%d %5s %5s %3s | %.4f %.4f | %s %s %s %s %s | %.3f %.3f | %.1f | %d %d %d' % ($P,'$1','$2','$EF',d['ms_per_step'],d['ms_per_step_sustained'],f('migrants'),f('halo_a'),f('halo_b'),f('one'),f('one_rest'),d['phases_ms']['dens'],d['phases_ms']['force'],d['host_wait_us']['mean'],pr['exchanges'],pr['one_message_steps'],pr['one_message_rests']))" >> $OUT
  done
  done
done
cat $OUT
