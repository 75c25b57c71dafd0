#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03o
export AB_LEN=1024 AB_STEPS=8
for rep in 1 2; do
  echo -n "bpermute: "; timeout -k 10 200 python tools/ab_fill.py
  echo -n "dpp     : "; BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/slim_dpp.so timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03o/ab_dpp.log
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_score_only.py -x -q -m gpu > gpurun_out/r03o/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03o/tests.log
