#!/bin/bash
# full GPU check: the -m gpu suite, smoke, the driver's bench line, kernel stats of the three models
TAG=${1:-r04d}; O=gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 2000 $O/bench.json
for cfg in "cls_b64:" "sem_seg_b16:--model sem_seg" "sem_seg_res_b8:--model sem_seg_res"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -f csv -- python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline --no-others --pipeline serial $flags > /dev/null 2>&1
  cp $O/prof_$name/p_kernel_stats.csv $O/${name}_kernel_stats.csv 2>/dev/null
  rm -rf $O/prof_$name
done
ls $O
