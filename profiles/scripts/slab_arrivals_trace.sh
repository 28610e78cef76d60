#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for L in default noinsert; do
  if [ "$L" = default ]; then unset SPH_HIP_LIB; else export SPH_HIP_LIB=$PWD/scratch/v/libsph_$L.so; fi
  rm -rf gpurun_out/prof_arr
  timeout -k 10 150 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_arr -o a -- python profiles/scripts/slab_arrivals_trace.py 100 > gpurun_out/arr_$L.log 2>&1
  echo "== $L"; tail -1 gpurun_out/arr_$L.log | cut -c1-300
  python - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/prof_arr/a_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("  kernel-busy total %.1f ms" % (tot/1e6))
for r in rows:
    if any(k in r["Name"] for k in ("k_mm_", "k_slab_", "k_cells", "k_os_", "copyBuffer")): print("   ", r["Name"][:44], r["Calls"], "avg %.1f us" % (float(r["AverageNs"])/1e3), "total %.2f ms" % (float(r["TotalDurationNs"])/1e6))
PY
done
rm -rf gpurun_out/prof_arr
