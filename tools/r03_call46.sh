#!/bin/bash
# counters of the final s=2 sweep on one config-4 chunk (low-half records), plus GRBM_GUI_ACTIVE for the issue fraction
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 bash tools/pmc_cfg4.sh r03v_cfg4 > gpurun_out/r03v_pmc.log 2>&1; echo "pmc rc=$?"
cd /tmp && export TMPDIR=/tmp CFG4_RUNS=2
timeout -k 10 200 rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r03v_cfg4 -o gui -- python3 $GRAFT_REPO_ROOT/tools/cfg4_chunk.py > /dev/null 2> $GRAFT_REPO_ROOT/gpurun_out/r03v_cfg4/gui.err; echo "gui rc=$?"
cd $GRAFT_REPO_ROOT && python tools/summarize_profile.py gpurun_out/r03v_cfg4 gpurun_out/r03v_cfg4/summary | tail -30
cat gpurun_out/r03v_cfg4/stats.log
