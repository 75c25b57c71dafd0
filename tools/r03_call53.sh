#!/bin/bash
# final tree: whole -m gpu suite, bench line, the other BASELINE shapes, the throughput matrix
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03x
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03x/gpu_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03x/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r03x/bench.json 2> gpurun_out/r03x/bench.err; echo "bench rc=$?"; cut -c1-400 gpurun_out/r03x/bench.json
timeout -k 10 900 python tools/perf_configs.py > gpurun_out/r03x/perf_configs.log 2>&1; echo "perf rc=$?"; cat gpurun_out/r03x/perf_configs.log
