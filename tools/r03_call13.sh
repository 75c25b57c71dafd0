#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
timeout -k 10 900 python -m pytest tests/test_gpu_packed_records.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/r03f/tests2.log 2>&1
echo "tests rc=$?"; tail -4 gpurun_out/r03f/tests2.log
export AB_LEN=1024 AB_STEPS=8
for rep in 1 2; do
  echo -n "new SLIM=1: "; timeout -k 10 200 python tools/ab_fill.py
  echo -n "OPT2=0 (old slim) SLIM=1: "; BIALIGN_LIB_OVERRIDE=$GRAFT_REPO_ROOT/build_exp/opt2_0.so timeout -k 10 200 python tools/ab_fill.py
done 2>&1 | tee gpurun_out/r03f/ab_salu.log
cd /tmp && export TMPDIR=/tmp AB_STEPS=3
out=$GRAFT_REPO_ROOT/gpurun_out/r03f/sq_slim_diet; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > $out/sq1.log 2> "$out/sq1.err"
