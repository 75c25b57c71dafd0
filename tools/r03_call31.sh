#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_dense_mu2.py -x -q -m gpu > gpurun_out/r03r/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03r/tests.log
export AB_LEN=1024 AB_STEPS=8
for rep in 1; do
  echo -n "low-half records: "; timeout -k 10 200 python tools/ab_fill.py
  echo -n "offset records  : "; BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/prev.so timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03r/ab.log
PERF_ONLY="cfg4,s=2,s=3" timeout -k 10 600 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03r/perf_configs.log
