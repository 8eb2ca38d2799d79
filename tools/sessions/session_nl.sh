#!/bin/bash
# nl_attention_direct: 4 waves per SIMD, scalar block addressing, permlane32_swap; parity + op timings
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reference_fixtures.py -x -q > $O/nl_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/nl_tests.log
timeout 300 python bench_ops.py --only nl 2>&1 | tee $O/nl_ops.log | grep -i "^nl"
