#!/bin/bash
# the next batch's prefix forked at the START of the step instead of at the head
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
run() { timeout 300 python bench.py --worker --model $1 $2 --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$PASNL_BENCH_PREFETCH_AT $1 $2', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))"; }
for at in head start; do export PASNL_BENCH_PREFETCH_AT=$at; run cls; run cls --AS; run sem_seg; run sem_seg_res; done
