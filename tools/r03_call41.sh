#!/bin/bash
# affine sweep at max_shift 0: ghost blocks of 8 (product) and 16 steps
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03s
for x in 16; do echo "== BLK $x"; BIALIGN_LIB_OVERRIDE=$PWD/exp_libs/blk$x.so timeout -k 10 300 python tools/s0_time.py 2>&1 | tee gpurun_out/r03s/s0_blk$x.log; done
