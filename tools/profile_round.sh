#!/bin/bash
# Profile of the bench workload (run on the GPU box):  tools/profile_round.sh <name> [bench args]
# -> gpurun_out/<name>/{stats,fetch,write}_*.csv ; copy the summaries into profiles/<name>/ afterwards.
# The program follows `--` directly (no env/bash hop: the profiler's preload has initialised the GPU by then).
set -e
name=$1; shift
out=$GRAFT_REPO_ROOT/gpurun_out/$name
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extra "$@" > "$out/bench_under_stats.json" 2> "$out/stats.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o fetch -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out" -o write -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extra "$@" > /dev/null 2> "$out/write.err"
ls "$out"
