#!/bin/bash
# wide-band path: structure-of-arrays ring (coalesced loads and stores)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout -k 10 600 python -m pytest tests/test_gpu_wide_band.py -x -q -m gpu > gpurun_out/r03r/wide_tests_soa.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03r/wide_tests_soa.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/wide_time.py 2>&1 | tee gpurun_out/r03r/wide_time_soa.log
BIALIGN_ALLOW_EXPERIMENT_BUILD=8 BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/exp8.so timeout -k 10 200 python tools/wide_phases.py 2>&1 | tee gpurun_out/r03r/wide_phases_soa.log
