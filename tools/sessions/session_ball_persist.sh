#!/bin/bash
O=gpurun_out/r04q; mkdir -p $O; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ref_kernels.py tests/test_gpu_properties.py -x -q -k "ball or query" > $O/ball_tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/ball_tests.log
for i in 1 2; do
python tools/ball_ab.py 2>&1 | grep -v amdgpu
python tools/ball_ab.py pointasnl_amd/csrc/libpasnl_hip_bgnp.so 2>&1 | grep -v amdgpu
done
