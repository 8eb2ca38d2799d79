#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reference_fixtures.py -x -q -k "wide or sem_seg_res or seg_res or layer" > $O/wide128_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/wide128_tests.log
run() { timeout 300 python bench.py --worker --model sem_seg_res --steps 20 --warmup 5 --no-cpu-baseline --no-others $1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'prefetch ms', d['ms_per_step'], 'serial', d['config'].get('serial_ms_per_step'), 'agree', d['config'].get('outputs_agree'))
for k in d['kernels']:
    if 'sa_cell' in k['kernel']: print('   ', k['kernel'], k['dims'], k.get('avg_us'), 'us', k.get('TFLOP/s'), 'TF')"; }
run ""; run "--set pointasnl_util.SA_CELL_SINGLE=0"; run ""; run "--set pointasnl_util.SA_CELL_SINGLE=0"
