#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py -x -q -m gpu > gpurun_out/r03d/packed_tests.log 2>&1
echo "packed tests rc=$?"; tail -8 gpurun_out/r03d/packed_tests.log
export AB_LEN=1024 AB_STEPS=8
for slim in 1 0; do
  echo -n "BIALIGN_SLIM=$slim len 1024: "; BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03d/ab_slim_len1024.log
export AB_LEN=512
for slim in 1 0; do
  echo -n "BIALIGN_SLIM=$slim len 512: "; BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03d/ab_slim_len512.log
