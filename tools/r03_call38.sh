#!/bin/bash
# wide-band path: per-part flag words instead of a shared counter, codes and score tables staged in LDS
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout -k 10 600 python -m pytest tests/test_gpu_wide_band.py -x -q -m gpu > gpurun_out/r03r/wide_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03r/wide_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python tools/wide_time.py 2>&1 | tee gpurun_out/r03r/wide_time.log
