#!/bin/bash
# A/B on one box: config-4 chunk with ghost ring (product), without (GSIDE=0), and ring without its stores (EXP=5)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
echo "== product (ghost ring)"; CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/ab_ring.log
echo "== GSIDE=0"; BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/gside0.so CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/ab_gside0.log
echo "== EXP=5 (no ghost stores; wrong results)"; BIALIGN_ALLOW_EXPERIMENT_BUILD=5 BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/exp5.so CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/ab_exp5.log
