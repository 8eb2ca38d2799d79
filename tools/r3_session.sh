#!/bin/bash
# round-3 GPU session: full GPU test suite + one replayed-forward timeline per model
out=gpurun_out/${1:-r3f}; mkdir -p $out
export TMPDIR=/tmp
( time timeout 1500 python -m pytest tests -x -q -m gpu > $out/gpu_tests.log 2>&1 ) 2> $out/gpu_tests.time; echo "gpu tests rc=$?"; tail -6 $out/gpu_tests.log; tail -3 $out/gpu_tests.time
for model in cls sem_seg sem_seg_res; do
  rm -rf $out/tl_$model
  timeout 300 rocprofv3 --kernel-trace -d $out/tl_$model -o tl -f csv -- python bench.py --worker --model $model --steps 6 --warmup 2 --no-cpu-baseline --no-others > $out/tl_$model.json 2> $out/tl_$model.err
  python tools/timeline.py $out/tl_$model 2 > $out/timeline_$model.txt 2>&1; tail -1 $out/timeline_$model.txt
done
