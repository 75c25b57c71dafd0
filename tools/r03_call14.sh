#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03g/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -6 gpurun_out/r03g/gpu_tests.log
python bench.py > gpurun_out/r03g/bench.json 2> gpurun_out/r03g/bench.err; echo "bench rc=$?"; cut -c1-1500 gpurun_out/r03g/bench.json
