#!/bin/bash
# SQ counters of the pair kernels for several library variants on ONE flowing C3 snapshot (GPU box, repo root):
#   [PMC_COUNTERS="FETCH_SIZE"] bash profiles/collect_pmc_ab.sh out_tag lib1.so lib2.so ...
# Clock-independent A/B: wall-clock A/B runs of one box differ by +-3 % (the chip sits at its power cap and the kernels
# of a step share one thermal budget), SQ_BUSY_CYCLES / SQ_WAVE_CYCLES / SQ_INSTS_* per launch do not.
set -e
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export KB_SNAP=/tmp/c3_flow.snap
[ -f $KB_SNAP ] || python profiles/scripts/kbench_flow.py --one prepare
OUT=gpurun_out/pmc_ab; rm -rf $OUT; mkdir -p $OUT
for SPEC in "$@"; do                       # lib.so or lib.so@SPH_BLOCK_ORDER=0,0,4 (an environment setting for that run)
  LIB=${SPEC%%@*}; SET=${SPEC#*@}; [ "$SET" = "$SPEC" ] && SET=""
  N=$(basename $LIB .so)${SET:+_$(echo $SET | tr -c 'A-Za-z0-9' '_')}
  export SPH_HIP_LIB=$(realpath $LIB)
  unset SPH_BLOCK_ORDER; [ -n "$SET" ] && export "$SET"
  rocprofv3 --pmc ${PMC_COUNTERS:-SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE} --kernel-include-regex 'k_force|k_density' \
      --output-format csv -d $OUT/$N -o p -- python profiles/scripts/kbench_flow.py --one run > $OUT/$N.log 2>&1
  python - "$OUT/$N" "$N" <<'PY' >> gpurun_out/${TAG}_pmc_ab.txt
import csv, glob, sys
from collections import defaultdict
root, name = sys.argv[1], sys.argv[2]
f = glob.glob(root + "/**/*counter_collection.csv", recursive=True)[0]
per = defaultdict(dict); kn = {}
for r in csv.DictReader(open(f, newline="")):
    d = int(r["Dispatch_Id"]); per[d][r["Counter_Name"]] = per[d].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    kn[d] = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("sph::", "")
for pat in ("k_density", "k_force<true, true, true"):
    ids = sorted(d for d in per if kn[d].startswith(pat))[10:]
    if not ids: continue
    m = {c: sum(per[d].get(c, 0.0) for d in ids) / len(ids) for c in per[ids[0]]}
    print(name, pat, len(ids), "launches:", " ".join(f"{c} {v:.4g}" for c, v in sorted(m.items())), flush=True)
PY
  rm -rf $OUT/$N
done
cat gpurun_out/${TAG}_pmc_ab.txt
