#!/bin/bash
# round 5: the classifier's first group_all module (one hand-written kernel) on a side stream
for i in 1 2; do
for f in "" layer1 head; do
  PASNL_BENCH_GA_FORK=$f timeout 200 python bench.py --steps 30 --warmup 5 --no-others --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('ga fork [$f]:', d['ms_per_step'], d['config'].get('serial_ms_per_step'), d['config'].get('outputs_agree'))"
done
done
