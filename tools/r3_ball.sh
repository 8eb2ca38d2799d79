#!/bin/bash
# ball-query iteration: parity, sweep, phase probe
out=gpurun_out/${1:-r3c}; mkdir -p $out
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ref_kernels.py tests/test_gpu_properties.py tests/test_gpu_cells.py -x -q -k "ball or repulsion" > $out/ball_tests.log 2>&1; echo "ball tests rc=$?"; tail -4 $out/ball_tests.log
timeout 600 python bench_ops.py --only ball --sweep --out $out/ball_sweep.json 2>&1 | grep -v amdgpu.ids | tee $out/ball_sweep.log | tail -8
python tools/ballprobe.py 2>&1 | grep -v amdgpu.ids | tee $out/ballprobe.txt
