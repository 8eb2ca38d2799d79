#!/bin/bash
# One-GPU rehearsal of the driver's multi-rank launch: torch.distributed.run (agent store) -> supervisor -> worker -> RCCL,
# once normally and once with the first worker stalled (the serial retry must rendezvous under the fresh key prefix).
mkdir -p gpurun_out/dist
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 \
  bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/dist/a.json 2> gpurun_out/dist/a.err
echo "normal rc=$?"; cut -c1-330 gpurun_out/dist/a.json
PASNL_BENCH_FAKE_STALL=run PASNL_BENCH_STALL_SCALE=0.1 timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 \
  --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 1 --force-dist --steps 20 --warmup 5 --no-cpu-baseline \
  > gpurun_out/dist/b.json 2> gpurun_out/dist/b.err
echo "stalled rc=$?"; cut -c1-330 gpurun_out/dist/b.json; grep -i "retry\|error\|Traceback" gpurun_out/dist/b.err | head -5
