#!/bin/bash
# s=3 packed sweep: low-half records (product) against offset records
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03u
for lib in "" exp_libs/lh1.so; do
  echo "== lib ${lib:-product}"
  for shape in "512 512" "128 1024" "1384 128"; do set -- $shape
    BIALIGN_LIB_OVERRIDE=${lib:+$PWD/$lib} CFG4_PAIRS=$1 CFG4_LEN=$2 CFG4_S=3 CFG4_RUNS=3 timeout -k 10 300 python tools/cfg4_chunk.py 2>&1 | tail -1
  done
done 2>&1 | tee gpurun_out/r03u/s3_lowhalf_ab.log
