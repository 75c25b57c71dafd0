#!/bin/bash
# fuzz soak on the final kernels, more multi-strip / team-capable shapes than the default draw
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
FUZZ_BIG=0.4 timeout -k 10 560 python tools/fuzz_gpu.py 500 71 > gpurun_out/r03z/fuzz_g.log 2>&1; echo "fuzz g rc=$?"; tail -1 gpurun_out/r03z/fuzz_g.log
FUZZ_BIG=0.15 timeout -k 10 460 python tools/fuzz_gpu.py 400 72 > gpurun_out/r03z/fuzz_h.log 2>&1; echo "fuzz h rc=$?"; tail -1 gpurun_out/r03z/fuzz_h.log
