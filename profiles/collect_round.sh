#!/bin/bash
# Everything the round's numbers come from, in one gpurun call (TAG names the files: r02_a, r02_b ...):
#   gpurun --timeout 1500 -- 'bash profiles/collect_round.sh r02_a'
# then here:  python profiles/pmc_to_traffic.py gpurun_out profiles/traffic.json ; cp gpurun_out/<TAG>_* profiles/
# Passes: bench lines (cls, cls --AS, sem_seg, sem_seg_res; default = serial pipeline), rocprofv3 --kernel-trace --stats of the
# same commands (worker form: profilers wrap the measurement process), PMC passes for HBM traffic (FETCH_SIZE / WRITE_SIZE in
# SEPARATE runs) and MFMA busy cycles, the per-op sweep.  Counter passes carry --kernel-trace + --pmc only.
TAG=${1:-r02}
export TMPDIR=/tmp
mkdir -p gpurun_out
O=gpurun_out
python bench.py --steps 50 --warmup 10 > $O/${TAG}_cls_b64_bench.json 2> $O/${TAG}_cls.err
python bench.py --steps 50 --warmup 10 --AS --no-cpu-baseline > $O/${TAG}_cls_AS_b64_bench.json 2>> $O/${TAG}_cls.err
python bench.py --steps 50 --warmup 10 --pipeline lanes --no-cpu-baseline > $O/${TAG}_cls_b64_lanes_bench.json 2>> $O/${TAG}_cls.err
python bench.py --model sem_seg --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_sem_seg_b16_bench.json 2>> $O/${TAG}_cls.err
python bench.py --model sem_seg_res --steps 20 --warmup 5 --no-cpu-baseline > $O/${TAG}_sem_seg_res_b8_bench.json 2>> $O/${TAG}_cls.err
for cfg in "cls_b64:" "cls_AS_b64:--AS" "sem_seg_b16:--model sem_seg" "sem_seg_res_b8:--model sem_seg_res"; do
  name=${cfg%%:*}; flags=${cfg#*:}
  rm -rf $O/prof_$name
  timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$name -o p -f csv -- python bench.py --worker --steps 20 --warmup 5 --no-cpu-baseline $flags > /dev/null 2>&1
  cp $O/prof_$name/p_kernel_stats.csv $O/${TAG}_${name}_kernel_stats.csv 2>/dev/null
done
# (HBM traffic: `bash profiles/collect_traffic.sh` in its OWN gpurun call BEFORE this one, then profiles/pmc_to_traffic.py here
# and a commit, so that the bench lines below find a traffic.json whose provenance matches the kernels they run)
bash profiles/collect_mfma_util.sh > $O/${TAG}_mfma_util.log 2>&1; cp $O/mfma_util.json $O/${TAG}_mfma_util.json
python bench_ops.py --sweep --out $O/${TAG}_bench_ops_sweep.json > $O/${TAG}_bench_ops.log 2>&1
ls $O | grep ${TAG} | head -30
