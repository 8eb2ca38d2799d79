#!/bin/bash
O=gpurun_out/r04f; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reference_fixtures.py tests/test_gpu_ops.py -x -q -k "sa_cell or seg or sem_seg or layer or fps" > $O/cells_tests.log 2>&1; echo "cells tests rc=$?"; tail -6 $O/cells_tests.log
for m in sem_seg_res sem_seg; do
timeout 600 python bench.py --worker --model $m --steps 20 --warmup 5 --no-cpu-baseline --no-others 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$m prefetch ms',d['ms_per_step'],'serial',d['config'].get('serial_ms_per_step'),'agree',d['config']['outputs_agree'])
for k in d['kernels'][:26]: print(f\"  {k['avg_us']:8.1f} us x{k['launches']:3d} {k['kernel']:28s} {k['dims']} {k['TFLOP/s']} TF {k['GB/s']} GB/s\")
" | tee $O/$m.txt
done
