#!/bin/bash
cd $GRAFT_REPO_ROOT
export AB_LEN=1024 AB_STEPS=3
cd /tmp && export TMPDIR=/tmp
for slim in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/r03e/sq_slim$slim
  mkdir -p $out
  export BIALIGN_SLIM=$slim
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > $out/sq1.log 2> "$out/sq1.err"
  rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$out" -o sq2 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/sq2.err"
  rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d "$out" -o sq3 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/sq3.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/write.err"
  cat $out/sq1.log
done
