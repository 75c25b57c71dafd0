#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/lds_rate tools/lds_rate.hip && timeout -k 10 200 /tmp/lds_rate > gpurun_out/r03c/lds_rate.txt 2>&1
cat gpurun_out/r03c/lds_rate.txt
echo "== one-layer, full records"; timeout -k 10 300 python tools/occupancy_probe.py 2>&1 | tee gpurun_out/r03c/occ_linear.log
echo "== one-layer, score only"; AB_SCORE_ONLY=1 timeout -k 10 300 python tools/occupancy_probe.py 2>&1 | tee gpurun_out/r03c/occ_linear_so.log
echo "== affine s=0 score only"; AB_AFFINE=1 AB_S=0 AB_SCORE_ONLY=1 timeout -k 10 300 python tools/occupancy_probe.py 2>&1 | tee gpurun_out/r03c/occ_affine_s0_so.log
