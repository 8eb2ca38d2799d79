#!/bin/bash
# per-forward launch counts of everything that is not a pasnl kernel (serial graph replays under rocprofv3)
export TMPDIR=/tmp; O=gpurun_out/glue; mkdir -p $O
for cfg in "sem_seg_res:--model sem_seg_res" "sem_seg:--model sem_seg" "cls:"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -f csv -- python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline --no-others --pipeline serial $flags > /dev/null 2>&1
  cp $O/prof_$name/p_kernel_stats.csv $O/${name}_serial_kernel_stats.csv
done
