#!/bin/bash
# probe: does a third wave per SIMD pay for the s=1 packed sweep?  (BLK 2 ring so that six 2-wave workgroups fit a CU at len 256)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
export AB_LEN=256 AB_STEPS=10 BIALIGN_TEAM=2
for pairs in 1024 1536 3072; do
  export AB_PAIRS=$pairs
  echo "== pairs $pairs"
  tools/ab_run.sh 2 build_exp/blk2.so build_exp/blk2w3.so || exit 1
done > gpurun_out/r03c/probe_3waves.log 2>&1
cat gpurun_out/r03c/probe_3waves.log
timeout -k 10 900 python -m pytest tests/test_gpu_dropin.py::test_config5_all_eight_shards_on_one_gpu -x -q -m gpu > gpurun_out/r03c/new_tests3.log 2>&1
echo "new tests rc=$?"; tail -5 gpurun_out/r03c/new_tests3.log
