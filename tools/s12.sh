#!/bin/bash
out=gpurun_out/s12; mkdir -p $out; export TMPDIR=/tmp
bash tools/gpu_session.sh s12 tests bench prof pmc seg
