#!/bin/bash
# s=2 sweep: eight-wave workgroups (diet layout), alone and as cross-CU teams, against the previous shapes.
cd $GRAFT_REPO_ROOT
export AB_CYCLES=1
for team in "" x16 h4 h2 8; do
  echo "== config-4 chunk (64 x 2000), BIALIGN_TEAM=$team"; BIALIGN_TEAM=$team timeout -k 10 200 python tools/ab_rna.py || exit 1
done
for team in "" 4 8; do
  echo "== 512 x 512 s=2, BIALIGN_TEAM=$team"; AB_PAIRS=512 AB_LEN=512 BIALIGN_TEAM=$team timeout -k 10 200 python tools/ab_rna.py || exit 1
done
for team in "" 4 8 h2; do
  echo "== 256 x 1000 s=2, BIALIGN_TEAM=$team"; AB_PAIRS=256 AB_LEN=1000 BIALIGN_TEAM=$team timeout -k 10 200 python tools/ab_rna.py || exit 1
done
