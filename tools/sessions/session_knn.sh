#!/bin/bash
O=gpurun_out/r04j; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_knn_grid.py -x -q -k nanoflann > $O/knn_tests.log 2>&1; echo "knn tests rc=$?"; tail -5 $O/knn_tests.log
timeout 600 python bench_ops.py --only knn_tree --iters 5 --out $O/knn_tree.json 2>&1 | grep -v amdgpu.ids | tee $O/knn_tree.log | tail -10
