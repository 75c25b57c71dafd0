#!/bin/bash
# Fill time for every (pairs, team) at one max_shift: tools/team_table.sh <s> <len> "<pairs list>" "<team list>"
s=$1; len=$2
for p in $3; do
  for t in $4; do
    echo -n "s=$s len=$len pairs=$p team=$t: "
    BIALIGN_TEAM=$t AB_AFFINE=${AB_AFFINE:-1} AB_LEAN=${AB_LEAN:-0} AB_PAIRS=$p AB_LEN=$len AB_S=$s AB_CYCLES=1 AB_RUNS=4 timeout -k 10 200 python tools/ab_alloc.py | head -1 | sed 's/cycle 0: fill ms //' || exit 1
  done
done
