#!/bin/bash
# Composition of the s=2 cross-CU sweep (one chunk of config 4): team size, hand-off waits, stores.
# Timing builds (BIALIGN_EXP) compute wrong results by construction.  Run on the GPU box.
set -e
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp_s2
tools/exp_build.sh $PWD/gpurun_out/exp_s2/nowait.so BIALIGN_EXP=9 > /dev/null
tools/exp_build.sh $PWD/gpurun_out/exp_s2/nostore.so BIALIGN_EXP=1 > /dev/null
export AB_CYCLES=1
for team in x20 x16 x12 x10 x8; do
  echo "== default build, BIALIGN_TEAM=$team"; BIALIGN_TEAM=$team timeout -k 10 200 python tools/ab_rna.py
done
echo "== no hand-off waits (EXP 9)"; BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/nowait.so timeout -k 10 200 python tools/ab_rna.py
echo "== no stores (EXP 1)"; BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/nostore.so timeout -k 10 200 python tools/ab_rna.py
echo "== no stores, x16"; BIALIGN_TEAM=x16 BIALIGN_LIB_OVERRIDE=$PWD/gpurun_out/exp_s2/nostore.so timeout -k 10 200 python tools/ab_rna.py
echo "== 128 pairs x 1000 (in-workgroup team 4 vs cross-CU)"
AB_PAIRS=256 AB_LEN=1000 timeout -k 10 200 python tools/ab_rna.py
AB_PAIRS=256 AB_LEN=1000 BIALIGN_TEAM=4 timeout -k 10 200 python tools/ab_rna.py
AB_PAIRS=256 AB_LEN=1000 BIALIGN_TEAM=x8 timeout -k 10 200 python tools/ab_rna.py
rm -f gpurun_out/exp_s2/*.so; rm -rf gpurun_out/exp_s2/*.obj
