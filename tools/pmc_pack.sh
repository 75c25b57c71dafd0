#!/bin/bash
# Dynamic instruction counts of the config-2 sweep with packed and with full records (run on the GPU box).
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export AB_STEPS=3
for pk in 0 1; do
  export BIALIGN_PACK=$pk
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o sq_pack$pk -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > "$out/sq_pack$pk.log" 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write_pack$pk -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch_pack$pk -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2>&1
done
