#!/bin/bash
# HBM traffic of the hand-written kernels from PMC counters (MI355X_MICROARCH.md "HBM"): FETCH_SIZE and WRITE_SIZE
# in SEPARATE rocprofv3 passes (TCC slot limit), kernel-trace only.  Run on the GPU box through gpurun:
#   gpurun -- 'bash profiles/collect_traffic.sh'   -> gpurun_out/pmc_{fetch,write}/..., then profiles/pmc_to_traffic.py
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d gpurun_out/pmc_$c -o cls -f csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-graph > gpurun_out/pmc_$c.json 2> gpurun_out/pmc_$c.err || true
done
find gpurun_out -name "*counter_collection.csv" | head
