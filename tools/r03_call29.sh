#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/pmc_cfg4.sh r03b_cfg4 > gpurun_out/r03b_cfg4.log 2>&1 || { tail -5 gpurun_out/r03b_cfg4.log; exit 1; }
tail -3 gpurun_out/r03b_cfg4/stats.log
