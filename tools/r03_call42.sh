#!/bin/bash
# final tree of round 3 (after the wide-band rewrite): the whole -m gpu suite, then the default bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03t
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03t/gpu_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r03t/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python bench.py > gpurun_out/r03t/bench.json 2> gpurun_out/r03t/bench.err; echo "bench rc=$?"; cat gpurun_out/r03t/bench.json
