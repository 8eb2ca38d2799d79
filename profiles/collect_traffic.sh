#!/bin/bash
# HBM traffic of the hand-written kernels from PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 passes (TCC slot limit), kernel-trace only.  Run on the GPU box through gpurun:
#   gpurun -- 'bash profiles/collect_traffic.sh [bench args]'  -> gpurun_out/pmc_{FETCH,WRITE}_SIZE/, then
#   python profiles/pmc_to_traffic.py gpurun_out profiles/traffic.json   (here)
export TMPDIR=/tmp
mkdir -p gpurun_out
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$c
  timeout 600 rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o cls -f csv -- \
    python bench.py --worker --steps 3 --warmup 1 --no-cpu-baseline --no-graph --launch-order "$@" > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err || true
done
find gpurun_out -name "*counter_collection.csv" | head
