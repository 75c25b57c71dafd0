#!/bin/bash
# last fuzz soak of round 3 (final tree), two seeds
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03end
timeout -k 10 560 python tools/fuzz_gpu.py 500 81 > gpurun_out/r03end/fuzz_i.log 2>&1; echo "fuzz i rc=$?"; tail -1 gpurun_out/r03end/fuzz_i.log
FUZZ_BIG=0.3 timeout -k 10 460 python tools/fuzz_gpu.py 400 82 > gpurun_out/r03end/fuzz_j.log 2>&1; echo "fuzz j rc=$?"; tail -1 gpurun_out/r03end/fuzz_j.log
