#!/bin/bash
# round 5: sa_tail with packed weights -- parity, then the models A/B on one box
timeout 600 python -m pytest tests/test_gpu_cells.py -q -x -k "sa_tail" 2>&1 | tail -3
timeout 600 python -m pytest tests/test_gpu_reference_fixtures.py -q -x -k "set_abstraction or model_matches" 2>&1 | tail -3
for i in 1 2; do
for m in cls sem_seg_res; do
  for sw in "" "--set pointasnl_util.SA_TAIL_PACKED=False"; do
      timeout 200 python bench.py --model $m --steps 30 --warmup 5 --no-others --no-cpu-baseline $sw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', '[$sw]', d['ms_per_step'], d['config'].get('serial_ms_per_step'), d['config'].get('outputs_agree'))"
  done
done
done
