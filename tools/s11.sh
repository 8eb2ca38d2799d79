#!/bin/bash
out=gpurun_out/s11; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py -m gpu -x -q -k "sa_cell or cls_forward" > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log | cut -c1-300
for cfg in 8 4; do
  echo "== sa_cell nw $cfg"; PASNL_SA_CELL_CFG=$cfg timeout 300 python bench_ops.py --only sacell --out $out/ops_sacell_$cfg.json 2>&1 | grep sa_
done
