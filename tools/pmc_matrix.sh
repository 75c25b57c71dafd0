#!/bin/bash
# Kernel stats and HBM counters of the kernels the bench does not exercise: affine s=0 / s=3, the one-layer (non-affine)
# sweep, score-only and lean-traceback storage.  usage (GPU box): tools/pmc_matrix.sh <name>
set -e
name=$1
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
export PERF_ONLY="s=3,s=0,non-affine,score-only,lean"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $GRAFT_REPO_ROOT/tools/perf_configs.py > "$out/stats.log" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/tools/perf_configs.py > /dev/null 2> "$out/write.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $GRAFT_REPO_ROOT/tools/perf_configs.py > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d "$out" -o sq -- python3 $GRAFT_REPO_ROOT/tools/perf_configs.py > /dev/null 2> "$out/sq.err"
cat "$out/stats.log"
