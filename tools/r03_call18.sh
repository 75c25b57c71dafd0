#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
PERF_ONLY="cfg4,s=2,s=3" timeout -k 10 900 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03h/perf_configs2.log
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_dense_mu2.py tests/test_gpu_xcu_residency.py -x -q -m gpu > gpurun_out/r03h/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03h/tests.log
