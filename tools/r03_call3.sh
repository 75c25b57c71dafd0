#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rate tools/valu_rate.hip && VALU_RATE_MAX3=1 timeout -k 10 400 /tmp/valu_rate > gpurun_out/r03b/valu_rate2.txt 2>&1
cat gpurun_out/r03b/valu_rate2.txt
timeout -k 10 900 python -m pytest tests/test_gpu_dropin.py::test_config5_all_eight_shards_on_one_gpu -x -q -m gpu > gpurun_out/r03b/new_tests2.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r03b/new_tests2.log
