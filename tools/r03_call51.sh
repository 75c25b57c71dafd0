#!/bin/bash
# 300 and 384 pairs x len 1024: which team shape is best when eight- or twelve-wave workgroups need a second round
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
{
for n in 300 384; do for team in "" 6 12 8 4 x6 x5 x4; do
  echo -n "pairs $n BIALIGN_TEAM=$team: "; BIALIGN_TEAM=$team AB_PAIRS=$n AB_LEN=1024 AB_STEPS=6 timeout -k 10 200 python tools/ab_fill.py 2>&1 | tail -1
done; done
} 2>&1 | tee gpurun_out/r03w/teams_300.log
