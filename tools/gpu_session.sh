#!/bin/bash
# One gpurun call = tests + bench + rocprofv3 kernel stats (+ optional extras).  Outputs under gpurun_out/<tag>/.
# usage: gpurun -- 'bash tools/gpu_session.sh <tag> [tests] [bench] [prof] [seg] [ops] [pmc]'
tag=${1:-s}; shift
what=${*:-tests bench prof}
out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for w in $what; do
case $w in
tests) timeout 900 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $out/pytest.log;;
bench) timeout 600 python bench.py > $out/bench.json 2> $out/bench.err; echo "bench rc=$?"; cut -c1-600 $out/bench.json;
       timeout 600 python bench.py --AS --no-cpu-baseline > $out/bench_AS.json 2> $out/bench_AS.err; cut -c1-300 $out/bench_AS.json;
       timeout 600 python bench.py --pipeline serial --no-cpu-baseline > $out/bench_p1.json 2> $out/bench_p1.err; cut -c1-300 $out/bench_p1.json;;
prof)  timeout 900 rocprofv3 --kernel-trace --stats -d $out/prof -o cls -- python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline > $out/prof_bench.json 2> $out/prof.err
       python profiles/summarize_rocpd.py $out/prof/cls_results.db $out/kernel_stats.csv; head -40 $out/kernel_stats.csv;;
seg)   timeout 600 python bench.py --model sem_seg --steps 10 --warmup 3 > $out/bench_sem_seg.json 2> $out/bench_sem_seg.err; cut -c1-300 $out/bench_sem_seg.json
       timeout 600 python bench.py --model sem_seg_res --steps 10 --warmup 3 > $out/bench_sem_seg_res.json 2> $out/bench_sem_seg_res.err; cut -c1-300 $out/bench_sem_seg_res.json;;
ops)   timeout 900 python bench_ops.py --sweep --out $out/bench_ops.json > $out/bench_ops.log 2>&1; tail -60 $out/bench_ops.log;;
pmc)   bash profiles/collect_traffic.sh; cp gpurun_out/pmc_FETCH_SIZE.json gpurun_out/pmc_WRITE_SIZE.json $out/ 2>/dev/null;;
esac
done
