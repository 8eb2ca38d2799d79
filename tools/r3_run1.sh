#!/bin/bash
# round-3 GPU session 1: new ball-query kernel (parity + sweep), new bench line
out=gpurun_out/r3a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ref_kernels.py tests/test_gpu_properties.py -x -q -k "ball" > $out/ball_tests.log 2>&1; echo "ball tests rc=$?"; tail -5 $out/ball_tests.log
timeout 600 python bench_ops.py --only ball --sweep --out $out/ball_sweep.json > $out/ball_sweep.log 2>&1; echo "sweep rc=$?"; cat $out/ball_sweep.log | tail -12
( time timeout 900 python bench.py > $out/bench.json 2> $out/bench.err ) 2> $out/bench.time; echo "bench rc=$?"; tail -3 $out/bench.time; tail -5 $out/bench.err; cut -c1-400 $out/bench.json
python - <<'P'
import json
try:
    d=json.loads(open('gpurun_out/r3a/bench.json').read().strip().splitlines()[-1])
    print(d['value'], d['ms_per_step'], d['roofline'])
    for o in d['other_configs'] or []: print(o['workload'][:40], o['ms_per_step'], o['outputs_agree'], o['roofline'] and (o['roofline']['kernel'], o['roofline']['frac']))
    for s in d['ball_query_sweep'] or []: print(s)
    print(d['cpu_baseline'] and d['cpu_baseline']['value'])
except Exception as e: print('parse failed', e)
P
