#!/bin/bash
cd /tmp && export TMPDIR=/tmp AB_LEN=1024 AB_STEPS=3
out=$GRAFT_REPO_ROOT/gpurun_out/r03n/sq; mkdir -p $out
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_BRANCH --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > $out/sq1.log 2> "$out/sq1.err"
cat $out/sq1.log
