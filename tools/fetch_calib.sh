#!/bin/bash
# FETCH_SIZE against known byte counts in the ghost feed's access pattern (tools/fetch_calib.hip; run on the GPU box)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1
mkdir -p "$out"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/fetch_calib $GRAFT_REPO_ROOT/tools/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out" -o calib -- /tmp/fetch_calib > "$out/calib_known_bytes.txt" 2> "$out/calib.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o calibstats -- /tmp/fetch_calib > /dev/null 2>> "$out/calib.err"
cat "$out/calib_known_bytes.txt"
