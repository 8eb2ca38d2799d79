#!/bin/bash
# the four workloads, prefetch + serial, one line each
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
run() { timeout 300 python bench.py --worker --model $1 $2 --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))"; }
run cls; run cls --AS; run sem_seg; run sem_seg_res
