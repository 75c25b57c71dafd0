#!/bin/bash
# Counters of the s=2 cross-CU sweep on one chunk of config 4 (run on the GPU box): tools/pmc_cfg4.sh <name>
# Separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2; SQ: 8 per pass); program directly after `--`.
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export CFG4_RUNS=2
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > "$out/stats.log" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> "$out/write.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> "$out/sq1.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$out" -o sq2 -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> "$out/sq2.err"
ls "$out"
