#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03j
timeout -k 10 1100 python -m pytest tests/test_gpu_score_only.py tests/test_gpu_lean_trace.py tests/test_gpu_reduced_storage_oracle.py tests/test_gpu_packed_records.py -x -q -m gpu > gpurun_out/r03j/tests.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03j/tests.log
PERF_ONLY="cfg4,cfg5,score-only,lean" timeout -k 10 900 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03j/perf_configs.log
