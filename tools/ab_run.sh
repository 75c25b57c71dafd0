#!/bin/bash
# Alternating A/B of engine builds on the GPU box: tools/ab_run.sh <rounds> <a.so> <b.so> ...
# (process-to-process noise is ~0.7 ms at config 2, so variants are interleaved, several rounds)
rounds=$1; shift
for r in $(seq 1 $rounds); do
  for so in "$@"; do
    echo -n "$(basename $so) : "
    BIALIGN_LIB_OVERRIDE=$so timeout -k 10 120 python tools/ab_fill.py || exit 1
  done
done
