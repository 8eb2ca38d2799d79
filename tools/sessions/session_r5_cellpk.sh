#!/bin/bash
# round 5: the wide cells with packed weights -- parity, then the segmentation models A/B on one box
timeout 600 python -m pytest tests/test_gpu_cells.py -q -x -k "wide or single" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_reference_fixtures.py -q -x -k "model_matches or set_abstraction" 2>&1 | tail -3
for i in 1 2; do
for m in sem_seg_res sem_seg; do
  for sw in "" "--set pointasnl_util.SA_CELL_PACKED=False"; do
      timeout 200 python bench.py --model $m --steps 30 --warmup 5 --no-others --no-cpu-baseline $sw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', '[$sw]', d['ms_per_step'], d['config'].get('serial_ms_per_step'), d['config'].get('outputs_agree'))"
  done
done
done
