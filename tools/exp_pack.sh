#!/bin/bash
# Would packed records (base + 16-bit deltas in interior steps) pay?  Timing build BIALIGN_EXP=5
# (real encoding arithmetic, 4 instead of 7 stores per step, stand-in ghost decode; results wrong by construction)
# against the shipped build, alternating processes.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/exp_pack
tools/exp_build.sh $PWD/gpurun_out/exp_pack/pack.so BIALIGN_EXP=5 > /dev/null || exit 1
echo "== 1024 x 512 s=1"
AB_STEPS=8 tools/ab_run.sh 3 $PWD/bialign_amd/libbialign_hip.so $PWD/gpurun_out/exp_pack/pack.so
echo "== 1024 x 1024 s=1 (two chunks)"
AB_LEN=1024 AB_STEPS=5 tools/ab_run.sh 2 $PWD/bialign_amd/libbialign_hip.so $PWD/gpurun_out/exp_pack/pack.so
echo "== 512 x 512 s=2"
for so in $PWD/bialign_amd/libbialign_hip.so $PWD/gpurun_out/exp_pack/pack.so; do BIALIGN_LIB_OVERRIDE=$so AB_CYCLES=1 AB_PAIRS=512 AB_LEN=512 python tools/ab_rna.py; done
rm -rf gpurun_out/exp_pack/*.so gpurun_out/exp_pack/*.obj
