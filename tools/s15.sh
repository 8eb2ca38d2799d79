#!/bin/bash
out=gpurun_out/s15; mkdir -p $out; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_cells.py -m gpu -x -q 2>&1 | tail -8 | cut -c1-250
timeout 300 python bench.py --model sem_seg --steps 10 --warmup 3 > $out/bench_sem_seg.json 2> $out/e1; cut -c1-200 $out/bench_sem_seg.json; echo
timeout 300 python bench.py --model sem_seg_res --steps 10 --warmup 3 > $out/bench_sem_seg_res.json 2> $out/e2; cut -c1-200 $out/bench_sem_seg_res.json; echo
