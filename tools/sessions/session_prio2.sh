#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
run() { timeout 300 python tools/bench_with_lib.py $1 --worker --model $2 $3 --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2 $3', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))"; }
for lib in pointasnl_amd/csrc/libpasnl_hip_prio00.so pointasnl_amd/csrc/libpasnl_hip_prio03.so pointasnl_amd/csrc/libpasnl_hip_prio20.so; do
run $lib sem_seg_res; run $lib cls; run $lib cls --AS; run $lib sem_seg
done
