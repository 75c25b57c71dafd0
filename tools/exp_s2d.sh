#!/bin/bash
cd $GRAFT_REPO_ROOT
export AB_CYCLES=1
for team in "" x16 h4 h3; do
  echo "== config-4 chunk (64 x 2000), BIALIGN_TEAM=$team"; BIALIGN_TEAM=$team timeout -k 10 200 python tools/ab_rna.py || exit 1
done
echo "== 128 x 1400 s=2"; AB_PAIRS=128 AB_LEN=1400 timeout -k 10 200 python tools/ab_rna.py
echo "== 128 x 1400 s=2 x8"; AB_PAIRS=128 AB_LEN=1400 BIALIGN_TEAM=x8 timeout -k 10 200 python tools/ab_rna.py
