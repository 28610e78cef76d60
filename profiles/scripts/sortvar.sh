#!/bin/bash
# per-kernel stats of the full-sort step for several library variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for L in default "$@"; do
  if [ "$L" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/scratch/v/libsph_$L.so; fi
  rm -rf gpurun_out/prof_sv
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sv -o sv -- python profiles/scripts/fullsort_prof.py C3 > /dev/null 2>&1
  echo "== $L"
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_sv/sv_kernel_stats.csv")))
for r in rows:
    if any(k in r["Name"] for k in ("k_os_", "k_reorder", "k_cells")): print("  ", r["Name"][:48], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done
rm -rf gpurun_out/prof_sv
