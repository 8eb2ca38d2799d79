#!/bin/bash
O=gpurun_out/r04h; mkdir -p $O
timeout 300 tools/bf16x3_probe 2>&1 | tee $O/bf16x3_probe.txt
