#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip || exit 1
for k in "pair" "trio" "alone"; do VALU_RATE_ONLY="$k" VALU_RATE_MAX3=1 timeout -k 10 200 /tmp/valu_rate; done 2>&1 | tee gpurun_out/r03e/valu_rate4.txt
