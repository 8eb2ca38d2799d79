#!/bin/bash
# round 5: the bf16x3 product mode of the long GEMMs, end to end (same box, back to back)
mkdir -p gpurun_out/bx
for m in sem_seg_res sem_seg; do
  for sw in "" "--set tf_util.DENSE_BF16X3=True"; do
    for p in serial prefetch; do
      timeout 200 python bench.py --model $m --steps 20 --warmup 5 --no-others --no-cpu-baseline --pipeline $p $sw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', '$p', '[$sw]', d['ms_per_step'], d['config'].get('outputs_agree'))"
    done
  done
done
