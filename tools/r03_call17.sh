#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03h
timeout -k 10 900 python tools/perf_configs.py 2>&1 | tee gpurun_out/r03h/perf_configs.log
