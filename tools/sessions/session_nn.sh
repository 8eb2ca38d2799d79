#!/bin/bash
# three_nn rewrite: parity + op timings + the two segmentation models
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_properties.py tests/test_gpu_ref_kernels.py -x -q -k "three or interp" > $O/nn_tests.log 2>&1; echo "tests rc=$?"; tail -4 $O/nn_tests.log
timeout 300 python bench_ops.py --only interp 2>&1 | tee $O/nn_ops.log | grep -i "three_nn"
