#!/bin/bash
out=gpurun_out/s4; mkdir -p $out; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -15 $out/pytest.log | cut -c1-300
timeout 300 python bench.py --no-cpu-baseline > $out/bench.json 2> $out/bench.err; cut -c1-250 $out/bench.json; echo
PASNL_SA_CELL_V1=1 timeout 300 python bench.py --no-cpu-baseline > $out/bench_v1.json 2> $out/bench_v1.err; cut -c1-250 $out/bench_v1.json; echo
timeout 300 python bench_ops.py --only ball --sweep --out $out/ops_ball_grid.json > $out/ops_ball_grid.log 2>&1; tail -12 $out/ops_ball_grid.log
PASNL_BALL_BRUTE=1 timeout 300 python bench_ops.py --only ball --sweep --out $out/ops_ball_brute.json > $out/ops_ball_brute.log 2>&1; tail -12 $out/ops_ball_brute.log
