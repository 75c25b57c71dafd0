#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03m
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03m/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r03m/gpu_tests.log
