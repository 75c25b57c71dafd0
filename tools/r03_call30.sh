#!/bin/bash
cd $GRAFT_REPO_ROOT
tools/profile_headline.sh r03q/headline > gpurun_out/r03q_profile.log 2>&1 || { tail -5 gpurun_out/r03q_profile.log; exit 1; }
cut -c1-300 gpurun_out/r03q/headline/bench_under_stats.json
