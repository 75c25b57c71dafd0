#!/bin/bash
# round 3, GPU call 1: A/B of the prefetch variants at the headline shape, parity of the best candidate, baseline counters
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03a
export AB_LEN=1024 AB_STEPS=8
tools/ab_run.sh 2 bialign_amd/libbialign_hip.so build_exp/opt15.so build_exp/opt23.so build_exp/opt31.so > gpurun_out/r03a/ab_len1024.log 2>&1 || exit 1
cat gpurun_out/r03a/ab_len1024.log
BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/opt31.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_packed_records.py -x -q -m gpu > gpurun_out/r03a/parity_opt31.log 2>&1
echo "parity opt31 rc=$?"; tail -3 gpurun_out/r03a/parity_opt31.log
timeout -k 10 300 python -m pytest tests/test_gpu_bench.py -x -q -m gpu > gpurun_out/r03a/bench_tests.log 2>&1
echo "bench tests rc=$?"; tail -3 gpurun_out/r03a/bench_tests.log
tools/profile_headline.sh r03a/base > gpurun_out/r03a/profile.log 2>&1 || { tail -5 gpurun_out/r03a/profile.log; exit 1; }
tools/fetch_calib.sh r03a/calib
