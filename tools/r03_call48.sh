#!/bin/bash
# headline shape: three-waves-per-SIMD kernel (bpermute exchange) against the two-wave kernel with the lane-major exchange array
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03w
{
for slim in 1 0; do
  for shape in "1024 1024" "1024 512" "256 1024" "2048 512"; do set -- $shape
    echo -n "BIALIGN_SLIM=$slim $1 x $2: "; BIALIGN_SLIM=$slim AB_PAIRS=$1 AB_LEN=$2 AB_STEPS=6 timeout -k 10 300 python tools/ab_fill.py 2>&1 | tail -1
  done
done
} 2>&1 | tee gpurun_out/r03w/slim_vs_lanemajor.log
