#!/bin/bash
# round 5: the next batch's self-kNN as a background job (capped grid) -- parity, then the segmentation models over caps on one box
timeout 600 python -m pytest tests/test_gpu_knn_grid.py -q -x -k "background" 2>&1 | tail -3
for i in 1 2; do
for m in sem_seg_res sem_seg; do
  for w in 0 256 512 1024 2048; do
      PASNL_BENCH_PREFETCH_KNN_WGS=$w timeout 200 python bench.py --model $m --steps 30 --warmup 5 --no-others --no-cpu-baseline 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', 'cap $w:', d['ms_per_step'], d['config'].get('serial_ms_per_step'), d['config'].get('outputs_agree'))"
  done
done
done
