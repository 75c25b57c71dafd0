#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r03s/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -4 gpurun_out/r03s/gpu_tests.log
python bench.py > gpurun_out/r03s/bench.json 2> gpurun_out/r03s/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r03s/bench.json
tools/profile_headline.sh r03s/headline > gpurun_out/r03s/profile.log 2>&1 || { tail -5 gpurun_out/r03s/profile.log; exit 1; }
