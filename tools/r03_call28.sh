#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
FUZZ_BIG=0.35 timeout -k 10 420 python tools/fuzz_gpu.py 330 31 > gpurun_out/r03p/fuzz_a.log 2>&1; echo "fuzz a rc=$?"; tail -2 gpurun_out/r03p/fuzz_a.log
FUZZ_BIG=0.15 timeout -k 10 320 python tools/fuzz_gpu.py 240 32 > gpurun_out/r03p/fuzz_b.log 2>&1; echo "fuzz b rc=$?"; tail -2 gpurun_out/r03p/fuzz_b.log
