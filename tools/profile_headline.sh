#!/bin/bash
# Counter set of the headline launch (1024 protein pairs x len 1024, s=1, packed records; run on the GPU box):
#   tools/profile_headline.sh <name> [path of an engine build to profile instead of the shipped one]
# -> gpurun_out/<name>/: stats (bench.py itself), WRITE_SIZE, FETCH_SIZE, two SQ sets, GRBM_GUI_ACTIVE (tools/ab_fill.py at
# the same shape: same launch, no torch import).  Separate passes; the program follows `--` directly.
set -e
name=$1; shift
[ -n "$1" ] && export BIALIGN_LIB_OVERRIDE=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export AB_LEN=1024 AB_STEPS=3
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra > "$out/bench_under_stats.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > "$out/write.log" 2> "$out/write.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o sq1 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/sq1.err"
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d "$out" -o sq2 -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/sq2.err"
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d "$out" -o grbm -- python3 $GRAFT_REPO_ROOT/tools/ab_fill.py > /dev/null 2> "$out/grbm.err"
ls "$out"
