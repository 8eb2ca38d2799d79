#!/bin/bash
# counters of nl_attention_direct (one kernel, scannet shape), baseline library against the variant: bash tools/sessions/session_nlpmc.sh libA libB
export TMPDIR=/tmp
O=gpurun_out/r04q/nlpmc; rm -rf $O; mkdir -p $O
for lib in "$@"; do
  tag=$(basename $lib .so)
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
             "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
             "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_READ_sum" \
             "SQ_INST_LEVEL_VMEM SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL" \
             "SQ_VALU_MFMA_COEXEC_CYCLES SQ_LEVEL_WAVES SQ_IFETCH SQ_ACTIVE_INST_ANY"; do
    i=$((i+1))
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $O/${tag}_$i -o nl -f csv -- python tools/nl_workload.py $lib ${SHAPE:-scannet} > $O/${tag}_$i.log 2>&1 || echo "pass $i failed"
  done
done
python - <<'PY'
import csv, glob, collections, os
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/r04q/nlpmc/**/*counter_collection.csv", recursive=True):
    tag = f.split("nlpmc/")[1].split("/")[0].rsplit("_", 1)[0]
    for r in csv.DictReader(open(f)):
        if "nl_attention_direct" in r["Kernel_Name"]:
            acc[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for t in acc.values() for c in t})
tags = sorted(acc)
print(f"{'counter':40s}" + "".join(f"{t:>26s}" for t in tags))
for c in names:
    print(f"{c:40s}" + "".join(f"{(sum(acc[t][c]) / len(acc[t][c]) if acc[t][c] else float('nan')):26.1f}" for t in tags))
PY
