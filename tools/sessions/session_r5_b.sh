#!/bin/bash
# round 5, session 2: the whole GPU suite, the bench line, timelines of the replayed cls / sem_seg_res / sem_seg steps
O=gpurun_out/r05b; mkdir -p $O; export TMPDIR=/tmp
python -m pytest tests -q -x -m gpu 2>&1 | tail -5 | tee $O/pytest_gpu.txt
python bench.py --steps 50 --warmup 10 > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json
for M in cls sem_seg_res sem_seg; do
rm -rf $O/tl; PASNL_BENCH_TRACE_ONLY=1 timeout 300 rocprofv3 --kernel-trace -d $O/tl -o tl -f csv -- python bench.py --worker --model $M --steps 6 --warmup 2 --no-cpu-baseline --no-others > /dev/null 2>&1
MARK="sa_cell_kernel<64, 64, 8, false, true, true"; [ $M = sem_seg_res ] && MARK="sa_cell16_kernel"; [ $M = sem_seg ] && MARK="sa_cell_kernel<32, 32"
python tools/timeline.py $O/tl 2 "$MARK" > $O/timeline_${M}_prefetch.txt
rm -rf $O/tl
done
