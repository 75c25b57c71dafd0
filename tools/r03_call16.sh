#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03g/gpu_tests2.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r03g/gpu_tests2.log
AB_LEN=1024 AB_STEPS=8 timeout -k 10 200 python tools/ab_fill.py
