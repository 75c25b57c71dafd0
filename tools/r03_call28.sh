#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03p
FUZZ_BIG=0.35 timeout -k 10 420 python tools/fuzz_gpu.py 330 41 > gpurun_out/r03p/fuzz_c.log 2>&1; echo "fuzz a rc=$?"; tail -2 gpurun_out/r03p/fuzz_c.log
FUZZ_BIG=0.15 timeout -k 10 320 python tools/fuzz_gpu.py 240 42 > gpurun_out/r03p/fuzz_d.log 2>&1; echo "fuzz b rc=$?"; tail -2 gpurun_out/r03p/fuzz_d.log
