#!/bin/bash
# session 1: ball v2 parity + sweep + probe + counters; nanoflann kNN timing
out=gpurun_out/r04a; mkdir -p $out
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_ref_kernels.py tests/test_gpu_properties.py tests/test_gpu_cells.py -x -q -k "ball or repulsion" > $out/ball_tests.log 2>&1; echo "ball tests rc=$?"; tail -5 $out/ball_tests.log
timeout 600 python bench_ops.py --only ball --sweep --out $out/ball_sweep.json 2>&1 | grep -v amdgpu.ids | tee $out/ball_sweep.log | tail -12
timeout 300 python tools/ballprobe.py 2>&1 | grep -v amdgpu.ids | tee $out/ballprobe.txt
bash profiles/collect_ball_counters.sh > /dev/null 2>&1; python profiles/ball_counters_summary.py gpurun_out $out/ball_counters.json 2>&1 | tail -3
timeout 600 python bench_ops.py --only knn_tree --iters 5 --out $out/knn_tree.json 2>&1 | grep -v amdgpu.ids | tee $out/knn_tree.log | tail -10
