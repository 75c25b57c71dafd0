#!/bin/bash
# low-half records at every max_shift (offset format removed): the affected suites, then config 4 in full
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
timeout -k 10 1000 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py tests/test_gpu_dropin.py tests/test_gpu_stress.py -x -q -m gpu > gpurun_out/r03u/tests_lowhalf_all.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03u/tests_lowhalf_all.log
[ $rc -eq 0 ] || exit $rc
CFG4_PAIRS=256 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03u/cfg4_full_lowhalf.log
PERF_ONLY="cfg4" timeout -k 10 600 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03u/perf_configs_cfg4.log
