#!/bin/bash
# final tree: fuzz soak, config-4 counters, throughput matrix
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03x
timeout -k 10 500 python tools/fuzz_gpu.py 400 61 > gpurun_out/r03x/fuzz_f.log 2>&1; echo "fuzz rc=$?"; tail -2 gpurun_out/r03x/fuzz_f.log
timeout -k 10 400 bash tools/pmc_cfg4.sh r03x_cfg4 > gpurun_out/r03x/pmc.log 2>&1; echo "pmc rc=$?"
(cd /tmp && export TMPDIR=/tmp CFG4_RUNS=2 && timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03x_cfg4 -o gui -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r03x_cfg4/gui.err); echo "gui rc=$?"
python tools/summarize_profile.py gpurun_out/r03x_cfg4 gpurun_out/r03x_cfg4/summary > /dev/null; cat gpurun_out/r03x_cfg4/stats.log
timeout -k 10 900 python tools/perf_matrix.py > gpurun_out/r03x/perf_matrix.md 2> gpurun_out/r03x/perf_matrix.err; echo "matrix rc=$?"; tail -5 gpurun_out/r03x/perf_matrix.md
