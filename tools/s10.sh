#!/bin/bash
out=gpurun_out/s10; mkdir -p $out; export TMPDIR=/tmp
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_BUSY_CYCLES" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY"; do
  tag=$(echo $set | cut -c1-14 | tr ' ' '_')
  PASNL_SA_CELL_CFG=4 timeout 300 rocprofv3 --kernel-trace --pmc $set -d $out/pmc_$tag -o p -f csv -- python bench_ops.py --only sacell --iters 3 --out $out/x.json > $out/pmc_$tag.log 2>&1
  python - <<PY
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "sa_cell" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"].split("(")[0], r["Grid_Size"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: round(sum(x)/len(x)) for c,x in v.items()})
PY
done
