#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_headline.sh r03i/headline > gpurun_out/r03i_profile.log 2>&1 || { tail -5 gpurun_out/r03i_profile.log; exit 1; }
cat gpurun_out/r03i/headline/bench_under_stats.json | cut -c1-600
