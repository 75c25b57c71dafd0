#!/bin/bash
# s=2 cross-CU sweep, second look: what do shorter ghost blocks (BLK 2: half the ring) cost, and stores into cache (EXP 2)?
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp_s2
tools/exp_build.sh $PWD/gpurun_out/exp_s2/blk2.so BIALIGN_BLK_OVERRIDE=2 > /dev/null
tools/exp_build.sh $PWD/gpurun_out/exp_s2/cache.so BIALIGN_EXP=2 > /dev/null
export AB_CYCLES=1
echo "== default build"; timeout -k 10 200 python tools/ab_rna.py
echo "== BLK 2"; BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/blk2.so timeout -k 10 200 python tools/ab_rna.py
echo "== stores into cache (EXP 2)"; BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/cache.so timeout -k 10 200 python tools/ab_rna.py
echo "== default build, 512 x 512 s=2 (in-workgroup 4)"; AB_PAIRS=512 AB_LEN=512 timeout -k 10 200 python tools/ab_rna.py
echo "== BLK 2, 512 x 512 s=2"; AB_PAIRS=512 AB_LEN=512 BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/blk2.so timeout -k 10 200 python tools/ab_rna.py
rm -f gpurun_out/exp_s2/*.so; rm -rf gpurun_out/exp_s2/*.obj
