#!/bin/bash
# Timing-experiment builds of the engine (results are WRONG by construction; never shipped).
# usage: tools/exp_build.sh <BIALIGN_EXP value> <output .so>
set -e
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DBIALIGN_EXP=$1 -o "$2" bialign_amd/csrc/bialign_capi.hip
