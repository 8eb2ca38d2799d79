#!/bin/bash
out=gpurun_out/s7; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log | cut -c1-300
for cfg in 8,1 8,0 4,1 4,0; do
  echo "== sa_cell cfg $cfg"; PASNL_SA_CELL_CFG=$cfg timeout 300 python bench_ops.py --only sacell --out $out/ops_sacell_$cfg.json 2>&1 | grep sa_
done
timeout 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; cut -c1-250 $out/bench.json; echo
