#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03n
export AB_LEN=1024 AB_STEPS=8
for rep in 1 2; do echo -n "len 1024: "; timeout -k 10 200 python tools/ab_fill.py; done 2>&1 | tee gpurun_out/r03n/ab.log
PERF_ONLY="cfg4,s=2,s=3,score-only" timeout -k 10 900 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03n/perf_configs.log
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_score_only.py -x -q -m gpu > gpurun_out/r03n/tests.log 2>&1
echo "tests rc=$?"; tail -3 gpurun_out/r03n/tests.log
