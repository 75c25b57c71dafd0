#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
export AB_LEN=1024 AB_STEPS=8
for rep in 1 2; do
for slim in 1 0; do
  echo -n "OPT2=1 SLIM=$slim: "; BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
  echo -n "OPT2=0 SLIM=$slim: "; BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/opt2_0.so BIALIGN_SLIM=$slim timeout -k 10 200 python tools/ab_fill.py
done; done 2>&1 | tee gpurun_out/r03f/ab_opt2.log
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03f/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03f/tests.log
