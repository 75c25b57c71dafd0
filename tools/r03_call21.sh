#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03k
timeout -k 10 1000 python tools/perf_matrix.py > gpurun_out/r03k/perf_matrix.md 2>&1
tail -45 gpurun_out/r03k/perf_matrix.md
