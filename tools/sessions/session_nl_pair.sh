#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_cells.py tests/test_gpu_reference_fixtures.py -x -q -k "nl or non_local or attention or elementwise or model or cls" > $O/nlpair_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/nlpair_tests.log
timeout 600 python tools/nl_pair_ab.py 2>&1 | grep -v amdgpu.ids
