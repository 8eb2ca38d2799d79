#!/bin/bash
# HBM traffic of the hand-written kernels from PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 passes (TCC slot limit), kernel-trace only.  One process per counter runs EVERY workload of
# bench.py (configs[1..4], eager, two marked forwards each) and the ball-query sweep.  On the GPU box through gpurun:
#   gpurun -- 'bash profiles/collect_traffic.sh'  -> gpurun_out/pmc_{FETCH,WRITE}_SIZE/, then
#   python profiles/pmc_to_traffic.py gpurun_out profiles/traffic.json   (here)
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 900 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o all -f csv -- \
    python bench.py --worker --traffic-pass > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err || true
done
# third pass (round 6): vector-issue busy cycles and the kernel's cycles -> `valu_frac` of the issue-bound kernels
rm -rf gpurun_out/pmc_VALU
timeout 900 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE -d gpurun_out/pmc_VALU -o all -f csv -- \
  python bench.py --worker --traffic-pass > gpurun_out/pmc_VALU.json 2> gpurun_out/pmc_VALU.err || true
find gpurun_out -name "*counter_collection.csv" | head
