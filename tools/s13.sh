#!/bin/bash
out=gpurun_out/s13; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-300
timeout 300 python bench_ops.py --only sacell --out $out/ops_sacell.json 2>&1 | grep sa_
bash tools/gpu_session.sh s13 bench pmc
