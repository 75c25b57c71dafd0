#!/bin/bash
# ghost ring, HBM budget 0.90 of free: config 4 timing (one chunk of 128, the full 256), s=3 512 x 512
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_chunk_ring90.log
CFG4_PAIRS=256 CFG4_RUNS=2 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/cfg4_full_ring90.log
