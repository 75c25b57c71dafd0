#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/r03e/gpu_tests.log 2>&1
echo "gpu tests rc=$?"; tail -8 gpurun_out/r03e/gpu_tests.log
export AB_LEN=1024 AB_STEPS=8
for slim in 1 0; do
  echo -n "BIALIGN_SLIM=$slim len 1024: "; BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03e/ab_len1024.log
export AB_LEN=512
for slim in 1 0; do
  echo -n "BIALIGN_SLIM=$slim len 512: "; BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03e/ab_len512.log
