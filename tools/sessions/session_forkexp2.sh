#!/bin/bash
# fork variants of sem_seg_res in the default bench flow (prefetch run, then the serial comparison in the same process)
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
run() { timeout 300 python bench.py --worker --model sem_seg_res --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))"; }
for v in 0 2 3 0 3; do PASNL_EXP_FORK=$v run "fork=$v"; done
