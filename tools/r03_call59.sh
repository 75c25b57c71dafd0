#!/bin/bash
# the tree at the end of round 3: whole -m gpu suite, the default bench line, smoke()
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03end
timeout -k 10 1100 python -m pytest tests/ -x -q -m gpu > gpurun_out/r03end/gpu_tests.log 2>&1
rc=$?; echo "tests rc=$rc"; tail -2 gpurun_out/r03end/gpu_tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > gpurun_out/r03end/bench.json 2> gpurun_out/r03end/bench.err; echo "bench rc=$?"; cut -c1-330 gpurun_out/r03end/bench.json
