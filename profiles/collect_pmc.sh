#!/bin/bash
# PMC passes over the flowing-dam bench (run on the GPU box from the repo root):
#   bash profiles/collect_pmc.sh [runup] [steps] [warmup] [tag]
# One rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE cannot share a pass on gfx950; --pmc is never
# combined with a trace domain), restricted to the step's big kernels; profiles/pmc_summarize.py keeps the dispatches
# of the timed window and writes profiles/pmc_traffic.json + profiles/<tag>_c3_flow_sq_counters.json.
set -e
RUNUP=${1:-6000}; STEPS=${2:-20}; WARM=${3:-5}; TAG=${4:-r05}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_flow; rm -rf $OUT; mkdir -p $OUT
KRE='k_force|k_density|k_mm_move|k_cells_build|k_os_pass'
pass() {   # name, counters...
  local name=$1; shift
  rocprofv3 --pmc "$@" --kernel-include-regex "$KRE" --output-format csv -d $OUT/$name -o p -- \
      python bench.py --runup $RUNUP --steps $STEPS --warmup $WARM --no-cpu --no-pmc > $OUT/$name.log 2>&1
  echo "pass $name done: $(grep -c . $OUT/$name/*counter_collection.csv 2>/dev/null || echo 0) rows"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT
python profiles/pmc_summarize.py $OUT $RUNUP $WARM $STEPS $TAG
cp profiles/pmc_traffic.json profiles/${TAG}_c3_flow_sq_counters.json gpurun_out/
rm -f $OUT/*/*counter_collection.csv
