#!/bin/bash
# UTCL1 counters of the fill kernel over several allocation cycles (one process): do the slow
# allocations translate worse?  usage (on the GPU box): tools/pmc_alloc.sh <out dir under gpurun_out>
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
AB_CYCLES=6 rocprofv3 --kernel-trace --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum \
   -d "$out" -o pmc --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/ab_alloc.py
ls "$out"
