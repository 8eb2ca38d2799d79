#!/bin/bash
# PMC counters of the grid ball-query kernel (ball_grid.hip): occupancy, VALU / LDS issue, LDS bank conflicts.  Counter passes
# only (--kernel-trace + --pmc, nothing else); one small group per pass.
#   gpurun -- 'bash profiles/collect_ball_counters.sh'   then   python profiles/ball_counters_summary.py gpurun_out profiles/r02_ball_counters.json
export TMPDIR=/tmp
mkdir -p gpurun_out
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS" \
           "SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_STORE SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  rm -rf gpurun_out/ballpmc_$i
  timeout 300 rocprofv3 --kernel-trace --pmc $set -d gpurun_out/ballpmc_$i -o ball -f csv -- python profiles/ball_workload.py > gpurun_out/ballpmc_$i.log 2>&1 || true
done
find gpurun_out -name "*counter_collection.csv" | head
