#!/bin/bash
# ghost ring stores: never written through (EXP=6) / always written through (EXP=7); timing only
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03q
for x in 6 7; do
echo "== EXP=$x"; BIALIGN_ALLOW_EXPERIMENT_BUILD=$x BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/exp$x.so CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tee gpurun_out/r03q/ab_exp$x.log
done
