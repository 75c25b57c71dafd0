#!/bin/bash
# wide-band rewrite (SoA ring, flag barrier, staged codes): wide + drop-in suites, then a fuzz soak
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03r
timeout -k 10 900 python -m pytest tests/test_gpu_wide_band.py tests/test_gpu_dropin.py -x -q -m gpu > gpurun_out/r03r/tests_wide_dropin.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03r/tests_wide_dropin.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 420 python tools/fuzz_gpu.py 300 51 > gpurun_out/r03r/fuzz_e.log 2>&1; echo "fuzz rc=$?"; tail -3 gpurun_out/r03r/fuzz_e.log
