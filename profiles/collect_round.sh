#!/bin/bash
# Everything a round's numbers come from, in one gpurun call (TAG names the files: r03_e ...):
#   gpurun --timeout 1500 -- 'bash profiles/collect_round.sh r03_e'      then here:  cp gpurun_out/<TAG>_* profiles/
# (HBM traffic: `bash profiles/collect_traffic.sh` in its OWN gpurun call BEFORE this one, then profiles/pmc_to_traffic.py here
# and a commit, so that the bench line below finds a traffic.json whose provenance matches the kernels it runs.)
# Passes: the driver's bench line (all BASELINE configs + the ball-query sweep in one line), the same line with
# --pipeline serial, rocprofv3 --kernel-trace --stats of every model's forward (worker form: profilers wrap the measurement
# process), MFMA busy cycles, the ball-query counters, the per-op sweep.  Counter passes carry --kernel-trace + --pmc only.
TAG=${1:-r03}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
python bench.py --steps 50 --warmup 10 > $O/${TAG}_bench.json 2> $O/${TAG}_bench.err
python bench.py --steps 50 --warmup 10 --pipeline serial --no-cpu-baseline > $O/${TAG}_bench_serial.json 2>> $O/${TAG}_bench.err
for cfg in "cls_b64:" "cls_AS_b64:--AS" "sem_seg_b16:--model sem_seg" "sem_seg_res_b8:--model sem_seg_res"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -f csv -- python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline --no-others $flags > /dev/null 2>&1
  cp $O/prof_$name/p_kernel_stats.csv $O/${TAG}_${name}_kernel_stats.csv 2>/dev/null
done
bash profiles/collect_mfma_util.sh > $O/${TAG}_mfma_util.log 2>&1; cp $O/mfma_util.json $O/${TAG}_mfma_util.json 2>/dev/null
bash profiles/collect_ball_counters.sh > /dev/null 2>&1; python profiles/ball_counters_summary.py $O $O/${TAG}_ball_counters.json > /dev/null 2>&1
python bench_ops.py --sweep --out $O/${TAG}_bench_ops_sweep.json > $O/${TAG}_bench_ops.log 2>&1
ls $O | grep ${TAG} | head -30
