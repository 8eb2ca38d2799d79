#!/bin/bash
# round 5: fused feature-propagation head + residual in the tail: tests, then the segmentation models A/B on one box
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_cells.py -q -x -k "fp_interpolate or sa_tail or three_" 2>&1 | tail -3
timeout 900 python -m pytest tests/test_gpu_reference_fixtures.py -q -x 2>&1 | tail -3
for m in sem_seg_res sem_seg; do
  for sw in "" "--set pointnet_util.FP_HEAD_FUSED=False --set pointasnl_util.FP_HEAD_FUSED=False"; do
    for p in serial prefetch; do
      timeout 200 python bench.py --model $m --steps 20 --warmup 5 --no-others --no-cpu-baseline --pipeline $p $sw 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$m', '$p', '[$sw]', d['ms_per_step'], d['config'].get('outputs_agree'))"
    done
  done
done
