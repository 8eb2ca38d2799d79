#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reference_fixtures.py tests/test_gpu_overlap.py -x -q > $O/single_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/single_tests.log
run() { timeout 300 python bench.py --worker --model $1 --steps 20 --warmup 5 --no-cpu-baseline --no-others $2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1 $2', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))"; }
run sem_seg_res ""; run sem_seg_res "--set pointasnl_util.SA_CELL_SINGLE=0"; run sem_seg_res ""; run sem_seg_res "--set pointasnl_util.SA_CELL_SINGLE=0"
