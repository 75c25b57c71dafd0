#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03l
timeout -k 10 900 python -m pytest tests/test_gpu_wide_band.py -x -q -m gpu > gpurun_out/r03l/wide_tests.log 2>&1
echo "wide tests rc=$?"; tail -5 gpurun_out/r03l/wide_tests.log
timeout -k 10 300 python tools/wide_time.py 2>&1 | tee gpurun_out/r03l/wide_time.log
