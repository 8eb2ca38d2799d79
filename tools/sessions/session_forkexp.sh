#!/bin/bash
# why does the serial sem_seg_res forward run layer 1's sampler AFTER layer 0 instead of beside it?  fork variants + graph queue counts
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
run() { timeout 300 python bench.py --worker --model sem_seg_res --pipeline serial --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'serial ms', d['ms_per_step'], 'agree', d['config'].get('outputs_agree'))"; }
for v in 0 1 2 3; do PASNL_EXP_FORK=$v run "fork=$v"; done
for q in 1 2 3 6 8; do DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run "graph_queues=$q"; done
for v in 2 3; do for q in 2 3; do PASNL_EXP_FORK=$v DEBUG_HIP_FORCE_GRAPH_QUEUES=$q run "fork=$v graph_queues=$q"; done; done
